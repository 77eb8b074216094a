#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for q in 4 8 16; do for cap in 0 256; do
echo "== GPU_MAX_HW_QUEUES=$q, output launches capped at $cap workgroups"
MDC_TRACE_ENV=MDC_PIPE_TRACE_FINE=1,MDC_EXP_OUT_WGS=$cap,GPU_MAX_HW_QUEUES=$q timeout 600 python tools/reader_trace.py 256 3 2>&1 | grep -av amdgpu.ids | grep -a "chunks" | tail -1
MDC_TRACE_ENV=MDC_EXP_OUT_WGS=$cap,GPU_MAX_HW_QUEUES=$q timeout 600 python tools/reader_trace.py 256 20 batch 2>&1 | grep -a "READER_RATE /" | sed 's#/root/repo/oracle/_ref/##'
done; done
