#!/bin/bash
# the headline launch per library build (product / mono_dataset_code_amd/variants/libmdc_hip_<name>.so), one process each, buffers from the allocator, a given
# plan: LIBS="product lutrep16" PROBE_SHAPE=128x16:128 ROUNDS=3 bash tools/lib_ab.sh <tag>
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-lib_ab}
mkdir -p $OUT
for r in $(seq ${ROUNDS:-3}); do for l in ${LIBS:-product}; do
  lib=""; [ $l != product ] && lib="$GRAFT_REPO_ROOT/mono_dataset_code_amd/variants/libmdc_hip_$l.so"
  MDC_LIB_HIP=$lib timeout 300 python tools/placed_probe.py auto 1 ${FRAMES:-4096} 2>&1 | grep -a "PLACED\|rror\|fault" | sed "s/^PLACED/LIB $l ($PROBE_SHAPE):/" | cut -c1-200 >> $OUT/lib_ab.txt
done; done
cat $OUT/lib_ab.txt
