#!/bin/bash
# round 3, run 1: the whole GPU suite with the fixed threads test (driver style: -x), the native thread soak, baselines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_01; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep real $O/pytest.txt
( time timeout 900 bash tools/soak_threads.sh 2500 ) > $O/thread_soak.txt 2>&1; cat $O/thread_soak.txt | tail -8
for wl in fused pyramid; do
timeout 400 python bench.py --workload $wl --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null > $O/bench_$wl.json
python -c "
import json;d=json.loads(open('$O/bench_$wl.json').readline());r=d['roofline'];print('$wl', r['frames_per_launch'], r['frac'], r['kernel_ms'], r['same_box_mix_ceiling']['ms_median'], r['frac_of_same_box_mix_ceiling'], r['kernel'])"
done
