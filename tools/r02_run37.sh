#!/bin/bash
# read side / write side alone for the four kernels at 1024 frames
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r37; mkdir -p $O
V=$PWD/mono_dataset_code_amd/variants
for v in default skipload skipstore; do
  echo "== $v" >> $O/sides.txt
  if [ $v = default ]; then timeout 600 python tools/footprint_curve.py 1024 2>&1 | grep -v amdgpu.ids >> $O/sides.txt
  else MDC_LIB_HIP=$V/libmdc_hip_$v.so timeout 600 python tools/footprint_curve.py 1024 2>&1 | grep -v amdgpu.ids >> $O/sides.txt; fi
done
cat $O/sides.txt
