#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
MDC_TRACE_ENV=MDC_PIPE_TRACE_FINE=1 timeout 600 python tools/reader_trace.py 256 3 2>&1 | grep -av amdgpu.ids | grep -a chunks | tail -2
