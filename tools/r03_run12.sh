#!/bin/bash
# round 3, run 12: full GPU suite (driver style), kernel soak, thread soak, the three bench workloads
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_12; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1
timeout 900 python tools/soak.py 72 > $O/soak.txt 2>&1; tail -2 $O/soak.txt
timeout 600 bash tools/soak_threads.sh 1000 > $O/thread_soak.txt 2>&1; grep THREAD_SOAK $O/thread_soak.txt | cut -c1-200
for wl in fused pyramid unmap; do
timeout 400 python bench.py --workload $wl --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null > $O/bench_$wl.json
python -c "
import json;d=json.loads(open('$O/bench_$wl.json').readline());r=d['roofline'];print('$wl', r['frames_per_launch'], r['frac'], r['kernel_ms'], r['same_box_mix_ceiling']['ms_median'], r['frac_of_same_box_mix_ceiling'], r['kernel'], d['parity'])"
done
