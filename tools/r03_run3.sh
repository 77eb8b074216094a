#!/bin/bash
# round 3, run 3: what bounds the scale-1 remap?  read side alone / write side alone / both, direct and two-stage kernels
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_03; mkdir -p $O
V=mono_dataset_code_amd/variants
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_stage or SMALL or all_flags or tiled_and_gather or pyramid" ) > $O/pytest.txt 2>&1; grep -E "passed|failed|Error|error" $O/pytest.txt | tail -3
L=default,$V/libmdc_hip_skipstore.so,$V/libmdc_hip_skipload.so
timeout 600 python tools/exp.py --out 1280x1024 --frames 1024 --libs $L --two-stage 2,1 2>&1 | grep -v amdgpu.ids | tee $O/exp_1280_1024f.txt
timeout 600 python tools/exp.py --out 1280x1024 --frames 128 --libs $L --two-stage 2,1 --iters 20 2>&1 | grep -v amdgpu.ids | tee $O/exp_1280_128f.txt
timeout 600 python tools/exp.py --out 640x480 --frames 1024 --libs $L --two-stage 2 2>&1 | grep -v amdgpu.ids | tee $O/exp_640_1024f.txt
timeout 600 python tools/exp.py --out 1280x1024 --frames 1024 --two-stage 2,1 --nbuf 2,3 --fpb 8,32 --pyramid 0,1 2>&1 | grep -v amdgpu.ids | tee $O/exp_knobs.txt
