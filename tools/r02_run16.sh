#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run16; mkdir -p $O
timeout 900 python tools/reader_rate.py 256 > $O/reader_rate.txt 2>&1
grep "==\|READER_RATE\|failed\|not built" $O/reader_rate.txt
