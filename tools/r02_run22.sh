#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run22; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_random.py tests/test_gpu_debug.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 300 python tools/sweep.py --frames 4096 --rounds 5 --iters 6 --fpb 32 --cols 128,64 --rows 16,32 --pad 0,1 > $O/sweep.txt 2>&1
grep -n "passed\|failed\|Error\|assert" $O/pytest.log | head -20; cat $O/rc.txt; grep -v amdgpu $O/sweep.txt
