#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run29; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_debug.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python tools/sweep.py --frames 4096 --rounds 4 --iters 5 --fpb 32,64 --cols 128,64 --rows 16,32 --nbuf 2,3,4 > $O/sweep.txt 2>&1
grep -n "passed\|failed\|Error\|assert" $O/pytest.log | head; cat $O/rc.txt; grep -v amdgpu $O/sweep.txt
