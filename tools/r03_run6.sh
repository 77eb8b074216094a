#!/bin/bash
# round 3, run 6: write-pattern microbenchmark; strip kernel with 4 / 5 / 10 waves per workgroup; strip parity
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_06; mkdir -p $O
timeout 600 tools/bin/wpat 1280 1024 1024 > $O/wpat_1280.txt 2>&1; cat $O/wpat_1280.txt
timeout 600 tools/bin/wpat 640 480 4096 > $O/wpat_640.txt 2>&1; grep -E "linear|128 x 16|128 x 32|640 x" $O/wpat_640.txt
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "strip or two_stage" ) > $O/pytest.txt 2>&1; grep -E "passed|failed|Error|error" $O/pytest.txt | tail -3
V=mono_dataset_code_amd/variants
timeout 600 python tools/exp.py --out 1280x1024 --frames 1024 --libs default,$V/libmdc_hip_stripw5.so,$V/libmdc_hip_stripw10.so --two-stage 1 --pyramid 0,1 2>&1 | grep -v amdgpu.ids | tee $O/exp_stripw.txt
