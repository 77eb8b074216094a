#!/bin/bash
# round 3, run 34: zero-copy tests, reader rates, thread soak
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_34; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu -k "zero_copy or host or reader or dropin or thread" ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -8
timeout 300 python tools/zero_copy_rate.py 2>&1 | grep -v "amdgpu.ids\|^Input\|^Out\|resolution\|Reading\|Success" | head -8 > $O/zero_copy.txt; cat $O/zero_copy.txt
timeout 900 python tools/reader_rate.py 256 > $O/reader_rate.txt 2>&1; grep -v "amdgpu.ids" $O/reader_rate.txt | tail -30
