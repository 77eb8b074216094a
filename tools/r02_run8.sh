#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/pmc.sh r02_pmc_64x32 -- --frames 1024 --fpb 32 --cols 64 --rows 32 > /dev/null 2>&1
bash tools/pmc.sh r02_pmc_128x16 -- --frames 1024 --fpb 32 --cols 128 --rows 16 > /dev/null 2>&1
cat gpurun_out/r02_pmc_64x32/summary.txt gpurun_out/r02_pmc_128x16/summary.txt
