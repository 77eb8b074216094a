#!/bin/bash
cd $GRAFT_REPO_ROOT
for rm in fov affine affine128; do echo "== remap $rm"; timeout 300 python tools/sweep.py --frames 4096 --rounds 3 --iters 5 --fpb 32 --cols 64,128 --rows 32,16 --remap $rm 2>&1 | grep -v amdgpu; done
