#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python tools/exp.py --out 640x480 --frames 4096 --pyramid 0 --taper 1 --fpb 48,64,80,96,128 --cols 128 --rows 16,32 --rounds 4 --iters 4 2>&1 | grep -v amdgpu.ids
