#!/bin/bash
# row-sync experiment: neighbouring workgroups of a tile row throttled to within K frames of each other
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r33
V=mono_dataset_code_amd/variants
# parity of a sync build first (random remaps + all flags through the C ABI)
MDC_LIB_HIP=$PWD/$V/libmdc_hip_sync2.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "process_host_all_flags or batch or full_size" > gpurun_out/r33/pytest.txt 2>&1; tail -2 gpurun_out/r33/pytest.txt
for ord in 1 0; do
timeout 900 python tools/sweep.py --frames 4096 --rounds 4 --iters 5 --fpb 32 --cols 128 --rows 16 --order $ord --libs default,$V/libmdc_hip_sync1.so,$V/libmdc_hip_sync2.so,$V/libmdc_hip_sync4.so >> gpurun_out/r33/sweep.txt 2>&1
done
timeout 900 python tools/sweep.py --frames 4096 --rounds 4 --iters 5 --fpb 32 --cols 64 --rows 32 --order 1 --libs default,$V/libmdc_hip_sync1.so,$V/libmdc_hip_sync2.so >> gpurun_out/r33/sweep.txt 2>&1
grep -v amdgpu.ids gpurun_out/r33/sweep.txt
