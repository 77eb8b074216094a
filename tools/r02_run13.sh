#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run13; mkdir -p $O
timeout 900 python -m pytest tests/test_multi_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
bash tools/final_profiles.sh r02a > $O/final.log 2>&1
grep -n "passed\|failed\|Error\|assert" $O/pytest.log | head -20; cat $O/rc.txt; tail -30 $O/final.log
