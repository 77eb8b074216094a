#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for d in 1 2 3 4; do
echo "== guess = last 1/$d of the subsequence (1 = whole subsequence)"
for q in 90 98 75; do
Q=$q MDC_LIB_HIP=$PWD/mono_dataset_code_amd/variants/libmdc_hip_huffg$d.so timeout 300 python tools/huffman_rate.py 2>&1 | grep -v amdgpu.ids | grep "KB per\|n  64\|rounds" | cut -c1-150
done
done
