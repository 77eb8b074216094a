#!/bin/bash
# round 3, run 28: DSO one-call path with the gradients per prefetched chunk on the chunk's stream
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_28; mkdir -p $O
( time timeout 900 python -m pytest tests -x -q -m gpu -k "grad or dso or strip or pyramid" ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -5
for i in 1 2; do timeout 300 python tools/dso_rate.py 2>&1 | grep -v amdgpu.ids > $O/dso_rate_$i.txt; tail -9 $O/dso_rate_$i.txt; done
