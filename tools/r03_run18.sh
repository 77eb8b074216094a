#!/bin/bash
# round 3, run 18: the strip kernel's own prefetch (MDC_OPT_PREFETCH_DIST) -- parity, then A/B against the chunked prefetch
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_18; mkdir -p $O
timeout 500 python tools/exp.py --out 1280x1024 --frames 1024 --pyramid 1 --prefetch 0 --pfdist=-1,1,2,3,4,6,8,12,16,24 --rounds 3 --iters 4 > $O/exp_pfdist.txt 2>&1; cat $O/exp_pfdist.txt
timeout 400 python tools/exp.py --out 1280x1024 --frames 1024 --pyramid 0 --prefetch -1 --pfdist=-1,3,6,12 --rounds 3 --iters 4 > $O/exp_pfdist_base.txt 2>&1; cat $O/exp_pfdist_base.txt
