#!/bin/bash
# Long run of tests/native/thread_soak.cpp (SURVEY.md 8b "Threading"): 8 threads x N iterations on ONE mdc_ctx, own streams,
# small and full-size cameras.  usage: tools/soak_threads.sh [iterations per thread (small camera)] > profiles/rNN_thread_soak.txt
set -u
cd "$(dirname "$0")/.."
IT=${1:-2500}
python - <<'PY' > /tmp/soak_dirs.txt
import tempfile
from mono_dataset_code_amd import synth
a = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="soak_small_"), ("0.349153 0.436593 0.493140 0.499021 0.933271", "320 256", "0.4 0.53 0.5 0.5 0", "160 120"))
b = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="soak_full_"))
print(a); print(b)
PY
SMALL=$(sed -n 1p /tmp/soak_dirs.txt); FULL=$(sed -n 2p /tmp/soak_dirs.txt)
oracle/_ref/thread_soak "$SMALL" 8 "$IT" 3 2>&1 | grep -E "THREAD_SOAK|MISMATCH|failed"
oracle/_ref/thread_soak "$FULL" 8 $((IT / 10)) 4 2>&1 | grep -E "THREAD_SOAK|MISMATCH|failed"
oracle/_ref/thread_soak "$SMALL" 16 $((IT / 2)) 1 2>&1 | grep -E "THREAD_SOAK|MISMATCH|failed"
