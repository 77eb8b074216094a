#!/bin/bash
# full GPU suite + driver-style bench + smoke at HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r34; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_fused.json 2> $O/bench_fused.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
grep -n "passed\|failed" $O/pytest.log | tail -2; cat $O/rc.txt; cat $O/bench_fused.json; tail -1 $O/smoke.txt
