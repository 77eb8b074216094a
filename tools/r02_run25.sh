#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run25; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/rc.txt
grep -n "passed\|failed\|Error\|assert" $O/pytest.log | head; cat $O/rc.txt; cat $O/bench.json; tail -1 $O/smoke.txt
