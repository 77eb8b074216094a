#!/bin/bash
# round 3, run 7: decomposition of the scale-1 remap: which part of the kernel costs what (direct and strip kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_07; mkdir -p $O
V=mono_dataset_code_amd/variants
timeout 600 python tools/exp.py --out 1280x1024 --frames 1024 --libs default,$V/libmdc_hip_skipstore.so,$V/libmdc_hip_skipload.so,$V/libmdc_hip_fake2.so,$V/libmdc_hip_fake2_skipload.so --two-stage 2 2>&1 | grep -v amdgpu.ids | tee $O/exp_direct.txt
timeout 600 python tools/exp.py --out 1280x1024 --frames 1024 --libs default,$V/libmdc_hip_skipstore.so,$V/libmdc_hip_skipload.so,$V/libmdc_hip_strip_ns.so,$V/libmdc_hip_strip_nc.so,$V/libmdc_hip_strip_ns_nc.so,$V/libmdc_hip_strip_ns_nc_sl.so --two-stage 1 2>&1 | grep -v amdgpu.ids | tee $O/exp_strip.txt
timeout 600 python tools/exp.py --out 1280x1024 --frames 1024 --libs default,$V/libmdc_hip_strip_ns_nc_sl.so --two-stage 1 --nbuf 1,2 --fpb 8,64 2>&1 | grep -v amdgpu.ids | tee $O/exp_strip2.txt
