#!/bin/bash
# exact pyramid allowance: parity, soak, rate
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r41; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_debug.py -x -q -m gpu -k "pyramid or random or debug" > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1
timeout 900 python tools/soak.py 150 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
for i in 1 2; do
timeout 400 python bench.py --workload pyramid --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null > $O/pyr_$i.json
python -c "
import json;d=json.loads(open('$O/pyr_$i.json').readline());r=d['roofline'];print(r['frames_per_launch'], r['frac'], r['kernel_ms'], r['same_box_mix_ceiling']['ms_median'], r['frac_of_same_box_mix_ceiling'], r['kernel'])"
done
