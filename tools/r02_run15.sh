#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run15; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 900 python tools/reader_rate.py 256 > $O/reader_rate.txt 2>&1
grep -n "passed\|failed\|Error\|assert" $O/pytest.log | head -20; cat $O/rc.txt; grep "==\|READER_RATE\|failed\|not built" $O/reader_rate.txt
