#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for n in 256 512 768; do
for ch in 64 128 256; do
echo "== $n frames, reader chunk $ch"
MDC_EXP_READER_CHUNK=$ch MDC_RATE_KINDS=zip_jpg timeout 900 python tools/reader_rate.py $n 2>&1 | grep -v amdgpu.ids | grep "batch" | tail -1
done
done
