#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for fc in 0 32 16; do
echo "== first chunk $fc (0 = 64)"
MDC_TRACE_ENV=MDC_READER_LOOKAHEAD=64,MDC_EXP_FIRST_CHUNK=$fc timeout 600 python tools/reader_trace.py 1024 10 single 2>&1 | grep -a "READER_RATE /" | sed 's#/root/repo/oracle/_ref/##'
MDC_TRACE_ENV=MDC_EXP_FIRST_CHUNK=$fc timeout 600 python tools/reader_trace.py 1024 10 single 2>&1 | grep -a "READER_RATE /" | sed 's#/root/repo/oracle/_ref/##'
MDC_TRACE_ENV=MDC_EXP_FIRST_CHUNK=$fc timeout 600 python tools/reader_trace.py 256 20 batch 2>&1 | grep -a "READER_RATE /" | sed 's#/root/repo/oracle/_ref/##'
done; done
