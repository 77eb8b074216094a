#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp6; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
timeout 600 python tools/sweep.py --libs default --frames 1024 --rounds 5 --iters 10 --rows 32 --nbuf 2,3,4 --fpb 0,64 2>&1 | grep -v amdgpu.ids > $O/sweep32.txt
timeout 600 python tools/sweep.py --libs default --frames 1024 --rounds 5 --iters 10 --rows 60,64 --order 0,1 --nbuf 2,3 --fpb 0,64 2>&1 | grep -v amdgpu.ids > $O/sweep60.txt
timeout 600 python tools/sweep.py --libs default,mono_dataset_code_amd/variants/libmdc_hip_v1.so --frames 1024 --rounds 5 --iters 10 2>&1 | grep -v amdgpu.ids > $O/ab.txt
cat $O/pytest.txt $O/sweep32.txt $O/sweep60.txt $O/ab.txt
