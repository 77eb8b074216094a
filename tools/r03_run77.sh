#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for oc in 0 1; do
timeout 600 python -m pytest tests/test_reader.py -x -q > gpurun_out/t.log 2>&1; grep -aE "passed|failed|^E " gpurun_out/t.log | tail -5
echo "== results: $oc (0 = written in place by the kernel, 1 = staged on the device + one copy per run of contiguous images)"
MDC_TRACE_ENV=MDC_EXP_OUT_COPY=$oc timeout 600 python tools/reader_trace.py 256 3 2>&1 | grep -av amdgpu.ids | grep -a "chunks" | tail -1
MDC_TRACE_ENV=MDC_EXP_OUT_COPY=$oc timeout 600 python tools/reader_trace.py 256 20 batch 2>&1 | grep -a "READER_RATE /" | sed 's#/root/repo/oracle/_ref/##'
MDC_TRACE_ENV=MDC_EXP_OUT_COPY=$oc timeout 600 python tools/reader_trace.py 1024 10 batch 2>&1 | grep -a "READER_RATE /" | sed 's#/root/repo/oracle/_ref/##'
MDC_TRACE_ENV=MDC_EXP_OUT_COPY=$oc timeout 600 python tools/reader_trace.py 1024 10 single 2>&1 | grep -a "READER_RATE /" | sed 's#/root/repo/oracle/_ref/##'
done
