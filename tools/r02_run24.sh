#!/bin/bash
cd $GRAFT_REPO_ROOT
for wl in fused unmap pyramid; do timeout 300 python bench.py --steps 20 --warmup 5 --workload $wl --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$wl', d['config']['frames_per_gpu_per_step'], r['frac'], r['kernel_ms'], r.get('frac_of_same_box_mix_ceiling'), r['same_box_mix_ceiling']['ms_median'], r['same_box_mix_ceiling']['what'][-60:], d['parity'])"; done
