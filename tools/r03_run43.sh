#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_43; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_reader.py -x -q -m gpu --durations=6 ) > $O/pytest.txt 2>&1; grep -E "passed|failed|s call|s setup" $O/pytest.txt | tail -9
