#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run14; mkdir -p $O
timeout 300 python tools/sweep.py --frames 4096 --rounds 4 --iters 6 --fpb 24,32,48,64,96 --cols 128 --rows 16 --order 0,1 > $O/sweep.txt 2>&1
grep -v amdgpu $O/sweep.txt
