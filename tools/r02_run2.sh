#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run2; mkdir -p $O
timeout 300 tools/bin/hbm_mix > $O/hbm_mix.txt 2>&1
timeout 300 python tools/sweep.py --frames 1024 --rounds 4 --fpb 0,205,103,52 --rows 32 --sched 0,1 > $O/sweep_il.txt 2>&1
timeout 300 python tools/sweep.py --frames 1020 --rounds 4 --fpb 34,204 --rows 32 --sched 0,1 > $O/sweep_1020.txt 2>&1
grep -v "^write lin\|^read lin\|pieces" $O/hbm_mix.txt; cat $O/sweep_il.txt $O/sweep_1020.txt
