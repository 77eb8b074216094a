#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for la in 64 128 256; do echo "== getImage in order, 1024-frame sequence, 10 passes, lookahead at most $la"; MDC_TRACE_ENV=MDC_READER_LOOKAHEAD=$la timeout 600 python tools/reader_trace.py 1024 10 single 2>&1 | grep -a "READER_RATE reader\|READER_RATE /" | sed 's#/root/repo/oracle/_ref/##'; done
echo "== getImages, 1024 per call, 10 passes"; timeout 600 python tools/reader_trace.py 1024 10 batch 2>&1 | grep -a "READER_RATE /" | sed 's#/root/repo/oracle/_ref/##'
echo "== getImages, 256 per call, 20 passes"; timeout 600 python tools/reader_trace.py 256 20 batch 2>&1 | grep -a "READER_RATE /" | sed 's#/root/repo/oracle/_ref/##'
