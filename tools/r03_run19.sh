#!/bin/bash
# round 3, run 19: is it the prefetch or the chunking?  chunked launches with and without the prefetch launches
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_19; mkdir -p $O
for i in 1 2; do
timeout 500 python tools/exp.py --out 1280x1024 --frames 1024 --pyramid 1 --prefetch=-1,0,24,48,96 --rounds 3 --iters 4 > $O/exp_with_$i.txt 2>&1; cat $O/exp_with_$i.txt
MDC_EXP_NO_PREFETCH_KERNEL=1 timeout 500 python tools/exp.py --out 1280x1024 --frames 1024 --pyramid 1 --prefetch=-1,0,24,48,96 --rounds 3 --iters 4 > $O/exp_without_$i.txt 2>&1; cat $O/exp_without_$i.txt
done
