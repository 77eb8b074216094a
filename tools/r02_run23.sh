#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run23; mkdir -p $O
for f in 1024 4096; do timeout 300 python bench.py --steps 20 --warmup 5 --workload unmap --frames $f --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('unmap', d['config']['frames_per_gpu_per_step'], r['frac'], r['kernel_ms'], r.get('frac_of_same_box_mix_ceiling'))"; done
for fp in 8 16 32 64; do timeout 300 python bench.py --steps 20 --warmup 5 --workload unmap --frames 4096 --fpb $fp --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('unmap fpb $fp', r['frac'], r['kernel_ms'])"; done
