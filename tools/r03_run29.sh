#!/bin/bash
# round 3, run 29: DSO path with PLAIN (Infinity-Cache-allocating) base / level stores: do the gradients then read the levels from the cache?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_29; mkdir -p $O
for i in 1 2; do
timeout 300 python tools/dso_rate.py 2>&1 | grep -v amdgpu.ids > $O/dso_nt_$i.txt; sed -n 2,3p $O/dso_nt_$i.txt
MDC_LIB_HIP=$PWD/mono_dataset_code_amd/variants/libmdc_hip_aux0.so timeout 300 python tools/dso_rate.py 2>&1 | grep -v amdgpu.ids > $O/dso_plain_$i.txt; sed -n 2,9p $O/dso_plain_$i.txt
done
