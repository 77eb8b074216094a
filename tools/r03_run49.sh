#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_49; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_zz_threads.py -x -q -m gpu ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -8
timeout 900 bash tools/soak_threads.sh 1000 > $O/thread_soak.txt 2>&1; cat $O/thread_soak.txt
