#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_40; mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -8
timeout 900 python tools/reader_rate.py 256 2>&1 | grep -v amdgpu.ids > $O/reader_rate.txt; cat $O/reader_rate.txt
