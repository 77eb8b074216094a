#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_42; mkdir -p $O
for q in 90 50 98; do
Q=$q MDC_LIB_HIP=$PWD/mono_dataset_code_amd/variants/libmdc_hip_huffrounds.so timeout 600 python tools/huffman_rate.py 2>&1 | grep -v amdgpu.ids | grep "KB per\|n   1\|n  64\|n 256\|rounds"
done
