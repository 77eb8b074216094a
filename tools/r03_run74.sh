#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_reader.py -x -q > gpurun_out/t.log 2>&1; grep -aE "passed|failed|^E " gpurun_out/t.log | tail -5
MDC_TRACE_ENV=MDC_PIPE_TRACE_FINE=1 timeout 600 python tools/reader_trace.py 256 3 2>&1 | grep -av amdgpu.ids | grep -a "chunks\|drained" | tail -2
bash tools/r03_run71.sh
