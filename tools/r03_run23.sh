#!/bin/bash
# round 3, run 23: two-stream chunked strip path as the product default -- parity, bench pyramid x2 (and streams=1), soak, thread soak
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_23; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multi_gpu.py -x -q -m gpu -k "strip or pyramid or dso" ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -5
timeout 600 python tools/exp.py --out 1280x1024 --frames 1024 --pyramid 1 --prefetch 0 --streams 1,2 --rounds 4 --iters 4 2>&1 | grep -v amdgpu.ids > $O/exp_streams.txt; cat $O/exp_streams.txt
timeout 600 python tools/exp.py --out 1280x1024 --frames 4096 --pyramid 1 --prefetch 0 --streams 1,2 --rounds 3 --iters 2 2>&1 | grep -v amdgpu.ids > $O/exp_streams_4096.txt; cat $O/exp_streams_4096.txt
for i in 1 2; do
timeout 400 python bench.py --workload pyramid --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null > $O/bench_pyr_$i.json
python -c "
import json;d=json.loads(open('$O/bench_pyr_$i.json').readline());r=d['roofline'];print('pyramid', r['frames_per_launch'], r['frac'], r['kernel_ms'], r['same_box_mix_ceiling']['ms_median'], r['frac_of_same_box_mix_ceiling'], r['kernel'], r.get('launches_per_step'), d['parity'])"
done
timeout 600 python tools/soak.py 24 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
timeout 300 python tools/dso_rate.py 2>&1 | grep -v amdgpu.ids > $O/dso_rate.txt; tail -9 $O/dso_rate.txt
