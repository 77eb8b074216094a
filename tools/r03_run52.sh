#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_52; mkdir -p $O
( time timeout 600 python -m pytest tests/test_reader.py -x -q -m gpu -k "on_and_off" ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -12
