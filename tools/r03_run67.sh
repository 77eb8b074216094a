#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_67; mkdir -p $O
( time timeout 900 python -m pytest tests/test_reader.py tests/test_dropin.py -x -q -m gpu ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -8
MDC_RATE_KINDS=zip_jpg timeout 900 python tools/reader_rate.py 512 2>&1 | grep -v amdgpu.ids | grep "READER_RATE reader" | head -2
MDC_READER_LOOKAHEAD=0 MDC_RATE_KINDS=zip_jpg timeout 900 python tools/reader_rate.py 512 2>&1 | grep -v amdgpu.ids | grep "READER_RATE reader" | head -1
