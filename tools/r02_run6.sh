#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run6; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 300 python tools/sweep.py --libs default,mono_dataset_code_amd/variants/libmdc_hip_r01.so --frames 1024 --rounds 6 --iters 20 --fpb 32 --cols 64 --rows 32 --order 0 --nbuf 0 > $O/sweep_ab.txt 2>&1
timeout 300 python tools/sweep.py --frames 1024 --rounds 5 --iters 20 --fpb 8,16,24,32,48 --cols 64,128 --rows 32,16 --order 0 --nbuf 0 > $O/sweep_fpb.txt 2>&1
head -3 $O/pytest.log | cut -c1-300; grep -n "passed\|failed\|Error\|assert" $O/pytest.log | head; cat $O/rc.txt; cat $O/sweep_ab.txt $O/sweep_fpb.txt | grep -v amdgpu
