#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run17; mkdir -p $O
timeout 600 python tools/host_path_rate.py > $O/host_path.txt 2>&1
timeout 900 python tools/reader_rate.py 512 > $O/reader_rate.txt 2>&1
grep "frames/s" $O/host_path.txt; grep "==\|READER_RATE" $O/reader_rate.txt
