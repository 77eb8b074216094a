#!/bin/bash
# footprint curve: kernel time per frame vs frames per launch
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r35; mkdir -p $O; : > $O/curve.txt
for wl in pyramid fused; do
for frames in 128 192 256 288 320 384 448 512 768; do
    r=$(timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-ceiling --no-tune --steps 20 --warmup 5 --frames $frames 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['roofline']['frac'], d['roofline']['kernel_ms'])")
    echo "$wl frames $frames : $r" >> $O/curve.txt
done
done
cat $O/curve.txt
