#!/bin/bash
# round 3, run 14: one-call DSO preprocessing, new image formats on the GPU path, reader rates
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_14; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1
timeout 600 python tools/dso_rate.py 384 2>&1 | grep -v amdgpu.ids | tee $O/dso_rate.txt
timeout 900 python tools/reader_rate.py 512 > $O/reader_rate.txt 2>&1; grep "READER_RATE\|==" $O/reader_rate.txt | head -20
