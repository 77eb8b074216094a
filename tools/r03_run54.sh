#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_54; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tapered" ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -8
