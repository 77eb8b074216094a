#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_50; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v "amdgpu.ids\|^Input\|^Out\|resolution\|Reading\|Success" | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null > $O/bench_default.json; python -c "
import json;d=json.loads(open('$O/bench_default.json').readline());r=d['roofline'];print(d['value'], d['ms_per_step'], r['frac'], r['kernel_ms'], r['frac_of_same_box_mix_ceiling'], d['cpu_baseline']['value'], d.get('parity'))"
