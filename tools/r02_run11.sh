#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run11; mkdir -p $O
timeout 300 python tools/rate_undistort_f32.py > $O/rate_f32.txt 2>&1
grep -v amdgpu $O/rate_f32.txt | tail -12
