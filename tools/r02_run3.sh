#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run3; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 300 python tools/sweep.py --frames 1024 --rounds 4 --fpb 0 --cols 64,128 --rows 32,16 --order 0,3 --nbuf 0,2,3 > $O/sweep_shapes.txt 2>&1
timeout 300 python tools/sweep.py --frames 1024 --rounds 4 --fpb 16,24,32,48,64 --cols 128 --rows 32 --order 0 > $O/sweep_fpb128.txt 2>&1
head -3 $O/pytest.log | cut -c1-300; grep -n "passed\|failed\|Error\|assert" $O/pytest.log | head; cat $O/rc.txt; cat $O/sweep_shapes.txt $O/sweep_fpb128.txt | grep -v amdgpu
