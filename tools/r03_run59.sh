#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for ch in 32 64 96 128; do
echo "== MDC_EXP_STRM_CHUNK=$ch"
MDC_EXP_STRM_CHUNK=$ch MDC_RATE_KINDS=zip_jpg timeout 900 python tools/reader_rate.py 256 2>&1 | grep -v amdgpu.ids | grep "batch" | tail -1
done
