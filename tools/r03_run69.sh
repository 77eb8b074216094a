#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for n in 256 1024; do echo "== $n"; timeout 600 python tools/reader_trace.py $n 4 2>&1 | grep -v amdgpu.ids | tail -40; done
