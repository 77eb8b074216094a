#!/bin/bash
# LDS budget experiment: fewer LUT replicas (bank conflicts) in exchange for more window buffers / workgroups per CU
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r36; mkdir -p $O
V=mono_dataset_code_amd/variants
timeout 900 python tools/sweep.py --frames 4096 --rounds 4 --iters 5 --fpb 32 --cols 128 --rows 16 --nbuf 2,3,4 --libs default,$V/libmdc_hip_lut16.so,$V/libmdc_hip_lut8.so > $O/sweep.txt 2>&1
timeout 900 python tools/sweep.py --frames 4096 --rounds 4 --iters 5 --fpb 32 --cols 64 --rows 32 --nbuf 2,3,4 --libs default,$V/libmdc_hip_lut16.so,$V/libmdc_hip_lut8.so >> $O/sweep.txt 2>&1
grep -v amdgpu.ids $O/sweep.txt
