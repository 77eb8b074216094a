#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for n in 256 768 1400; do
echo "== $n frames"
MDC_RATE_KINDS=zip_jpg timeout 900 python tools/reader_rate.py $n 2>&1 | grep -v amdgpu.ids | grep "batch" | tail -1
done
