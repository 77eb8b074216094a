#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp10; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > $O/pytest.txt
timeout 600 tools/variants.sh "- aux0 aux16 aux17 aux18 aux19 aux3" --frames 1024 --rounds 6 --iters 10 > $O/aux.txt 2>&1
timeout 600 python tools/sweep.py --workload unmap --frames 512 --rounds 4 --iters 10 --fpb 0,4,8,16 2>&1 | grep -v amdgpu.ids > $O/unmap.txt
cat $O/pytest.txt $O/aux.txt $O/unmap.txt
