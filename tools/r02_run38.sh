#!/bin/bash
# pyramid workload in the steady state (1024 frames per launch): tile shapes / frames per workgroup / buffers
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r38; mkdir -p $O; : > $O/pyr.txt
for shape in "128 16" "64 32" "128 32" "64 64"; do
  set -- $shape
  for fpb in 0 32; do
   for nbuf in 0 2; do
    r=$(timeout 300 python bench.py --workload pyramid --frames 1024 --no-cpu-baseline --no-ceiling --steps 10 --warmup 3 --tile-cols $1 --tile-rows $2 --fpb $fpb --nbuf $nbuf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['kernel'])")
    echo "tile $1 x $2 fpb $fpb nbuf $nbuf : $r" >> $O/pyr.txt
   done
  done
done
cat $O/pyr.txt
