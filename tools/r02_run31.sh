#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r02c128x32_fused --tile-cols 128 --tile-rows 32 > gpurun_out/prof_a.log 2>&1
bash tools/profile_bench.sh r02c64x32_fused --tile-cols 64 --tile-rows 32 > gpurun_out/prof_b.log 2>&1
tail -3 gpurun_out/prof_a.log
