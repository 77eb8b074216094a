#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run30; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_multi_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('tuned', r['frac'], r['kernel_ms'], r['tile'], d['config']['plan'], r['frac_of_same_box_mix_ceiling'])"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tune 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('built-in', r['frac'], r['kernel_ms'], r['tile'], d['config']['plan'], r['frac_of_same_box_mix_ceiling'])"; done
grep -n "passed\|failed\|Error\|assert" $O/pytest.log | head; cat $O/rc.txt
