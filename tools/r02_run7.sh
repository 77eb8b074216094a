#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run7; mkdir -p $O
timeout 300 tools/bin/hbm_mix > $O/hbm_mix.txt 2>&1
grep "chunked\|tile64x32 read+write plain-ld nt-st\|prefetch, barrier\|mix lin" $O/hbm_mix.txt
