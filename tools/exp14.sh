#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp14; mkdir -p $O
timeout 900 python tools/sweep.py --frames 1024 --rounds 5 --iters 10 --rows 60 --order 0,1 --fpb 32,40,43,52,64,86,103,128 2>&1 | grep -v amdgpu.ids > $O/rows60.txt
timeout 900 python tools/sweep.py --frames 1024 --rounds 5 --iters 10 --rows 32,60,64 --order 0,1 --fpb 32,43,64 --nbuf 0 2>&1 | grep -v amdgpu.ids > $O/cmp.txt
cat $O/rows60.txt $O/cmp.txt
