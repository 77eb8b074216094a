#!/bin/bash
# round 3, run 33: zero copy in the host-pointer calls -- tests, A/B, host path rates with threads, reader rates
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_33; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu -k "host or reader or dropin or thread or pin or capi or abi" ) > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^E " $O/pytest.txt | head -8
timeout 300 python tools/zero_copy_rate.py 2>&1 | grep -v "amdgpu.ids\|^Input\|^Out\|resolution\|Reading\|Success" > $O/zero_copy.txt; cat $O/zero_copy.txt
timeout 600 python tools/host_path_rate.py 2>&1 | grep "frames/s" > $O/host_path.txt; cat $O/host_path.txt
