#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run12; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_fused.json 2> $O/bench_fused.err; echo "bench rc=$?" >> $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 --workload pyramid --no-cpu-baseline > $O/bench_pyr.json 2> $O/bench_pyr.err; echo "bench rc=$?" >> $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 --workload unmap --no-cpu-baseline > $O/bench_unmap.json 2> $O/bench_unmap.err; echo "bench rc=$?" >> $O/rc.txt
head -3 $O/pytest.log | cut -c1-300; grep -n "passed\|failed\|Error\|assert" $O/pytest.log | head -20; cat $O/rc.txt; cat $O/bench_fused.json $O/bench_pyr.json $O/bench_unmap.json; tail -5 $O/bench_fused.err
