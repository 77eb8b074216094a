#!/bin/bash
# round 3, run 10: full-width tiles (640 x 4, 320 x 8) of the direct kernel on the headline camera
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_10; mkdir -p $O
for shape in "128 16" "128 32" "640 4" "320 8" "128 16" "640 4"; do set -- $shape
timeout 300 python tools/exp.py --out 640x480 --frames 1024 --two-stage 2 --cols $1 --rows $2 --fpb 0,16,32,64 2>&1 | grep -v amdgpu.ids | grep -v "^lib\|^out" | tee -a $O/exp_shapes.txt
done
timeout 300 python tools/exp.py --out 640x480 --frames 4096 --two-stage 2 --cols 640 --rows 4 --fpb 32,64 --iters 3 2>&1 | grep -v amdgpu.ids | grep -v "^lib" | tee -a $O/exp_shapes.txt
timeout 300 python tools/exp.py --out 640x480 --frames 4096 --two-stage 2 --cols 128 --rows 32 --fpb 32,64 --iters 3 2>&1 | grep -v amdgpu.ids | grep -v "^lib" | tee -a $O/exp_shapes.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size or strip" 2>&1 | grep -E "passed|failed" | tail -1
