#!/bin/bash
# round 3, run 17: prefetched chunks -- parity, bench pyramid x3 (and without prefetch), soak
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_17; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multi_gpu.py -x -q -m gpu -k "strip or pyramid or dso" ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -5
for i in 1 2 3; do
timeout 400 python bench.py --workload pyramid --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null > $O/bench_pyr_$i.json
python -c "
import json;d=json.loads(open('$O/bench_pyr_$i.json').readline());r=d['roofline'];print('pyramid', r['frames_per_launch'], r['frac'], r['kernel_ms'], r['same_box_mix_ceiling']['ms_median'], r['frac_of_same_box_mix_ceiling'], r['kernel'], d['parity'])"
done
timeout 600 python tools/soak.py 24 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
