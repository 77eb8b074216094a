#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run10; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_fused.json 2> $O/bench_fused.err; echo "bench rc=$?" >> $O/rc.txt
timeout 300 python tools/rate_undistort_f32.py > $O/rate_f32.txt 2>&1
head -3 $O/pytest.log | cut -c1-300; grep -n "passed\|failed\|Error\|assert" $O/pytest.log | head; cat $O/rc.txt; cat $O/bench_fused.json; grep -v amdgpu $O/rate_f32.txt | tail
