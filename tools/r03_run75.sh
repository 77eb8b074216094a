#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for cap in 0 256 512; do
echo "== in-place output launches capped at $cap workgroups (0 = no cap)"
MDC_TRACE_ENV=MDC_PIPE_TRACE_FINE=1,MDC_EXP_OUT_WGS=$cap timeout 600 python tools/reader_trace.py 256 3 2>&1 | grep -av amdgpu.ids | grep -a "chunks" | tail -1
MDC_TRACE_ENV=MDC_EXP_OUT_WGS=$cap timeout 600 python tools/reader_trace.py 256 20 batch 2>&1 | grep -a "READER_RATE /" | sed 's#/root/repo/oracle/_ref/##'
MDC_TRACE_ENV=MDC_EXP_OUT_WGS=$cap timeout 600 python tools/reader_trace.py 1024 10 single 2>&1 | grep -a "READER_RATE /" | sed 's#/root/repo/oracle/_ref/##'
done
