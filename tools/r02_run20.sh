#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run20; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python tools/vcal_rate.py 200 6 > $O/vcal_rate.txt 2>&1
grep -n "passed\|failed\|Error\|assert" $O/pytest.log | head -20; cat $O/rc.txt; grep "GPU\|CPU\|Error\|error" $O/vcal_rate.txt
