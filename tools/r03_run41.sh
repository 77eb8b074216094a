#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_41; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_reader.py tests/test_dropin.py -x -q -m gpu ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -12
bash tools/r03_run39.sh 2>&1 | grep -A6 "stage 2"
MDC_RATE_KINDS=zip_jpg timeout 900 python tools/reader_rate.py 256 2>&1 | grep -v amdgpu.ids | tail -4
