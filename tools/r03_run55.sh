#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for zc in 0 2; do
echo "== MDC_ZERO_COPY=$zc"
MDC_ZERO_COPY=$zc MDC_RATE_KINDS=zip_jpg timeout 900 python tools/reader_rate.py 256 2>&1 | grep -v amdgpu.ids | grep "batch\|stage" | tail -6
done
