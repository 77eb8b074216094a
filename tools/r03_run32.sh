#!/bin/bash
# round 3, run 32: the fused kernel on page-locked HOST pointers (zero copy) against the copy path
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_32; mkdir -p $O
timeout 300 python tools/zero_copy_rate.py 2>&1 | grep -v "amdgpu.ids\|^Input\|^Out\|resolution\|Reading\|Success" > $O/zero_copy.txt; cat $O/zero_copy.txt
