#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp8; mkdir -p $O
echo "== 1024 frames" > $O/amort.txt
timeout 600 python tools/sweep.py --libs default --frames 1024 --rounds 5 --iters 10 --rows 32,60 --fpb 32,64 2>&1 | grep -v amdgpu.ids >> $O/amort.txt
echo "== 4096 frames (ms per launch = 4x frames)" >> $O/amort.txt
timeout 600 python tools/sweep.py --libs default --frames 4096 --rounds 4 --iters 4 --rows 32,60 --fpb 32,128,256 2>&1 | grep -v amdgpu.ids >> $O/amort.txt
cat $O/amort.txt
