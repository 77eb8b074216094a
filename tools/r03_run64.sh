#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
MDC_RATE_KINDS=zip_png,zip_jpg timeout 1500 python tools/reader_rate.py 768 2>&1 | grep -v amdgpu.ids | grep "READER_RATE reader\|==\|--"
