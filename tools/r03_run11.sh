#!/bin/bash
# round 3, run 11: exact chunk sets in the plan -> full GPU suite; full-width tiles on the headline camera
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_11; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1
for shape in "128 16" "640 4" "320 8" "128 32" "128 16" "640 4"; do set -- $shape
MDC_DEBUG_PLAN=1 timeout 300 python tools/exp.py --out 640x480 --frames 1024 --two-stage 2 --cols $1 --rows $2 --fpb 0,32,64 2>&1 | grep -v amdgpu.ids | grep -v "^lib\|^out" | tee -a $O/exp_shapes.txt
done
