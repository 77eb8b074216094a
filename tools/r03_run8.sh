#!/bin/bash
# round 3, run 8: prefetch depth of the strip kernel (bytes in flight), full kernel and read side alone
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_08; mkdir -p $O
V=mono_dataset_code_amd/variants
timeout 600 python tools/exp.py --out 1280x1024 --frames 1024 --libs default,$V/libmdc_hip_skipstore.so,$V/libmdc_hip_strip_ns_nc.so --two-stage 1 --nbuf 1,2,3,4 2>&1 | grep -v amdgpu.ids | tee $O/exp_depth.txt
timeout 600 python tools/exp.py --out 1280x1024 --frames 1024 --two-stage 1 --nbuf 2,3,4 --pyramid 1 2>&1 | grep -v amdgpu.ids | tee $O/exp_depth_pyr.txt
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "strip" ) > $O/pytest.txt 2>&1; grep -E "passed|failed|Error|error" $O/pytest.txt | tail -3
