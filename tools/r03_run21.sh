#!/bin/bash
# round 3, run 21: chunks alternating over 2..4 streams x chunk size x frames per workgroup
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_21; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 500 python tools/exp.py --out 1280x1024 --frames 1024 --pyramid 1 --prefetch=16,24,32,48 --fpb 0,4,8,16 --rounds 3 --iters 4 2>&1 | grep -v amdgpu.ids > $O/$tag.txt; echo "## $tag $*"; cat $O/$tag.txt; }
run streams1 MDC_X=0
run streams2 MDC_EXP_STREAMS=2
run streams3 MDC_EXP_STREAMS=3
run streams4 MDC_EXP_STREAMS=4
