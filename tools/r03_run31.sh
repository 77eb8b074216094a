#!/bin/bash
# round 3, run 31: strip path WITHOUT the levels (1280x1024 base only) in prefetched chunks over two streams
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_31; mkdir -p $O
timeout 600 python tools/exp.py --out 1280x1024 --frames 1024 --pyramid 0 --prefetch=-1,24,32,40,48,64 --streams 1,2 --rounds 3 --iters 4 2>&1 | grep -v amdgpu.ids > $O/base_only.txt; cat $O/base_only.txt
timeout 600 python tools/exp.py --out 1280x1024 --frames 1024 --pyramid 0 --prefetch=32,40,48 --streams 2 --fpb 8,12,16,20,24 --rounds 3 --iters 4 2>&1 | grep -v amdgpu.ids > $O/base_only_fpb.txt; cat $O/base_only_fpb.txt
