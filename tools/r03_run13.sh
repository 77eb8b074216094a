#!/bin/bash
# round 3, run 13: after the locking / slots refactor and the bench-library split: whole suite, soaks, host path rates
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_13; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1
timeout 900 python tools/soak.py 36 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
timeout 600 bash tools/soak_threads.sh 2500 > $O/thread_soak.txt 2>&1; grep THREAD_SOAK $O/thread_soak.txt | cut -c1-220
timeout 600 python tools/host_path_rate.py 2> $O/host_path.txt >/dev/null; grep "frames/s" $O/host_path.txt
