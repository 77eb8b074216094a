#!/bin/bash
# round 3, run 20: chunks alternating over two streams (tails overlap the next chunk), prefetch order
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_20; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 500 python tools/exp.py --out 1280x1024 --frames 1024 --pyramid 1 --prefetch=0,16,24,32,48 --rounds 3 --iters 4 2>&1 | grep -v amdgpu.ids > $O/$tag.txt; echo "## $tag $*"; cat $O/$tag.txt; }
run one_stream MDC_X=0
run one_stream_pf_right_before MDC_EXP_PF_ORDER=1
run two_streams MDC_EXP_STREAMS=2
run two_streams_first_half MDC_EXP_STREAMS=2 MDC_EXP_FIRST=12
run two_streams_fpb8 MDC_EXP_STREAMS=2 MDC_EXP_FIRST=12
