#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_70; mkdir -p $O
( time timeout 900 python -m pytest tests/test_reader.py tests/test_dropin.py -x -q -m gpu ) > $O/pytest.txt 2>&1; grep -aE "passed|failed" $O/pytest.txt | tail -1; grep -aE "^E " $O/pytest.txt | head -8
for n in 256 1024; do echo "== $n"; timeout 600 python tools/reader_trace.py $n 4 2>&1 | grep -v amdgpu.ids | tail -7; done
MDC_RATE_KINDS=zip_jpg timeout 900 python tools/reader_rate.py 256 2>&1 | grep -v amdgpu.ids | grep -a "READER_RATE reader\|^--" | tail -7
