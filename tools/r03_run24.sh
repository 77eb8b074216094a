#!/bin/bash
# round 3, run 24: the headline (direct kernel, 1280x1024 -> 640x480) in prefetched chunks over one / two streams
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_24; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 600 python tools/exp.py --out 640x480 --frames 4096 --pyramid 0 --fpb 0,8,16,32 --rounds 3 --iters 3 2>&1 | grep -v amdgpu.ids > $O/$tag.txt; echo "## $tag $*"; cat $O/$tag.txt; }
run baseline MDC_X=0
for ch in 32 64 128 256; do
run chunk${ch}_2s MDC_EXP_TILED_CHUNK=$ch
run chunk${ch}_1s MDC_EXP_TILED_CHUNK=$ch MDC_EXP_TILED_STREAMS=1
done
run chunk64_2s_nopf MDC_EXP_TILED_CHUNK=64 MDC_EXP_TILED_NOPF=1
run chunk128_2s_nopf MDC_EXP_TILED_CHUNK=128 MDC_EXP_TILED_NOPF=1
