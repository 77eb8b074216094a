#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_66; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multi_gpu.py -x -q -m gpu -k "tune or bench or multi" ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -8
for i in 1 2; do
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null > $O/bench_$i.json; python -c "
import json;d=json.loads(open('$O/bench_$i.json').readline());r=d['roofline'];print(d['value'], r['frac'], r['kernel_ms'], r['frac_of_same_box_mix_ceiling'], d['config']['plan'])"
timeout 900 python bench.py --no-cpu-baseline --fpb 64 --tile-cols 128 --tile-rows 32 2>/dev/null > $O/bench_fpb64_$i.json; python -c "
import json;d=json.loads(open('$O/bench_fpb64_$i.json').readline());r=d['roofline'];print('  fixed 128x32 fpb 64:', r['frac'], r['kernel_ms'])"
done
