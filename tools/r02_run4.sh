#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 300 python tools/sweep.py --frames 1024 --rounds 5 --fpb 32,64 --cols 64,128 --rows 32,16 --order 0 --nbuf 0 > $O/sweep_shapes.txt 2>&1
timeout 300 python tools/sweep.py --frames 4096 --rounds 3 --iters 4 --fpb 32,64 --cols 64,128 --rows 32 --order 0 --nbuf 0,3 > $O/sweep_4096.txt 2>&1
head -3 $O/pytest.log | cut -c1-300; grep -n "passed\|failed\|Error\|assert" $O/pytest.log | head; cat $O/rc.txt; cat $O/sweep_shapes.txt $O/sweep_4096.txt | grep -v amdgpu
