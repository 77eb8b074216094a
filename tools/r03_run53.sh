#!/bin/bash
# round 3, run 53: tapered tail of large tiled launches -- parity first, then A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_53; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -x -q -m gpu -k "full_size or random or beyond or batch" ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -8
timeout 600 python tools/exp.py --out 640x480 --frames 4096 --pyramid 0 --taper 2,1 --fpb 0,32,48,64 --cols 128 --rows 16,32 --rounds 4 --iters 4 2>&1 | grep -v amdgpu.ids > $O/taper_headline.txt; cat $O/taper_headline.txt
timeout 600 python tools/exp.py --out 640x480 --frames 1024,  --pyramid 0 --taper 2,1 --rounds 2 --iters 4 2>&1 | tail -3
