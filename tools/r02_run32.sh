#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r32
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "process_host_all_flags or full_size" > gpurun_out/r32/pytest.txt 2>&1; tail -3 gpurun_out/r32/pytest.txt
timeout 600 python tools/sweep.py --frames 4096 --rounds 4 --iters 5 --fpb 32,64 --cols 128,256 --rows 16 > gpurun_out/r32/sweep.txt 2>&1
timeout 600 python tools/sweep.py --frames 4096 --rounds 4 --iters 5 --fpb 32,64 --cols 128 --rows 32 >> gpurun_out/r32/sweep.txt 2>&1
cat gpurun_out/r32/sweep.txt
