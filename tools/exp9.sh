#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp9; mkdir -p $O
MDC_LIB_HIP=$GRAFT_REPO_ROOT/mono_dataset_code_amd/variants/libmdc_hip_unmaplane.so timeout 600 python -m pytest tests -m gpu -x -q -k "unmap or all_flags" 2>&1 | tail -4 > $O/pytest_variant.txt
timeout 600 python tools/sweep.py --libs default,mono_dataset_code_amd/variants/libmdc_hip_unmaplane.so --workload unmap --frames 512 --rounds 5 --iters 10 --fpb 0,8,16,32 2>&1 | grep -v amdgpu.ids > $O/unmap.txt
cat $O/pytest_variant.txt $O/unmap.txt
