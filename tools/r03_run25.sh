#!/bin/bash
# round 3, run 25: the headline in chunks WITHOUT prefetch over 1 / 2 / 3 streams
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_25; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 600 python tools/exp.py --out 640x480 --frames 4096 --pyramid 0 --fpb 16,24,32,48,64 --rounds 3 --iters 3 2>&1 | grep -v amdgpu.ids > $O/$tag.txt; echo "## $tag $*"; cat $O/$tag.txt; }
run baseline MDC_X=0
for ch in 96 128 192 256 384 512; do
run chunk${ch}_2s MDC_EXP_TILED_CHUNK=$ch MDC_EXP_TILED_NOPF=1
done
for ch in 128 256; do
run chunk${ch}_1s MDC_EXP_TILED_CHUNK=$ch MDC_EXP_TILED_NOPF=1 MDC_EXP_TILED_STREAMS=1
run chunk${ch}_3s MDC_EXP_TILED_CHUNK=$ch MDC_EXP_TILED_NOPF=1 MDC_EXP_TILED_STREAMS=3
done
run baseline_again MDC_X=0
