#!/bin/bash
# round 3, run 60: long soaks with the round's final kernels (thread contract, tile shapes, chunked strip path, pool-sharing)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_60; mkdir -p $O
timeout 900 bash tools/soak_threads.sh 2500 > $O/thread_soak.txt 2>&1; cat $O/thread_soak.txt
D=$(python -c "
import tempfile
from mono_dataset_code_amd import synth
print(synth.write_sequence_calibration(tempfile.mkdtemp(prefix='soak_up_'), ('0.349153 0.436593 0.493140 0.499021 0.933271', '160 128', 'full', '640 512')))" 2>/dev/null | tail -1)
timeout 600 oracle/_ref/thread_soak $D 12 400 5 1 2>&1 | grep -E "THREAD_SOAK|MISMATCH|failed" | tee -a $O/thread_soak.txt
timeout 900 python tools/soak.py 90 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
