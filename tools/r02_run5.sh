#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run5; mkdir -p $O
echo skip > $O/pytest.log; echo "pytest skipped" > $O/rc.txt
timeout 300 python tools/sweep.py --libs default,mono_dataset_code_amd/variants/libmdc_hip_r01.so --frames 1024 --rounds 5 --fpb 32 --cols 64,128 --rows 32 --order 0 --nbuf 0 > $O/sweep_ab.txt 2>&1
head -3 $O/pytest.log | cut -c1-300; grep -n "passed\|failed\|Error\|assert" $O/pytest.log | head; cat $O/rc.txt; cat $O/sweep_ab.txt | grep -v amdgpu
