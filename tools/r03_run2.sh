#!/bin/bash
# round 3, run 2: two-stage kernel -- parity, soak, pyramid bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_02; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_stage or pyramid or tiled_and_gather" ) > $O/pytest.txt 2>&1; grep -E "passed|failed|Error|error" $O/pytest.txt | tail -3
timeout 600 python tools/soak.py 48 > $O/soak.txt 2>&1; tail -2 $O/soak.txt
b() { # name, args...
  n=$1; shift
  timeout 400 python bench.py --workload pyramid --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>$O/bench_$n.err > $O/bench_$n.json
  python -c "
import json;d=json.loads(open('$O/bench_$n.json').readline());r=d['roofline'];print('$n', r['frames_per_launch'], r['frac'], r['kernel_ms'], r['same_box_mix_ceiling']['ms_median'], r['frac_of_same_box_mix_ceiling'], r['kernel'], d['parity'])" || tail -3 $O/bench_$n.err
}
b direct --two-stage 2
b auto
b conv_nbuf3 --nbuf 3
b conv_64x32 --tile-cols 64 --tile-rows 32
b conv_fpb16 --fpb 16
b conv_fpb64 --fpb 64
b direct2 --two-stage 2
b auto2
