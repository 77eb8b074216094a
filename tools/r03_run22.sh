#!/bin/bash
# round 3, run 22: two streams, finer sweep of chunk size x frames per workgroup (+ staggered first chunk)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_22; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 900 python tools/exp.py --out 1280x1024 --frames 1024 --pyramid 1 --prefetch=24,28,32,36,40,48,56,64 --fpb 10,12,14,16,20,24,32 --rounds 3 --iters 4 2>&1 | grep -v amdgpu.ids > $O/$tag.txt; echo "## $tag $*"; cat $O/$tag.txt; }
run streams2 MDC_EXP_STREAMS=2
run streams2_first16 MDC_EXP_STREAMS=2 MDC_EXP_FIRST=16
run streams3 MDC_EXP_STREAMS=3
