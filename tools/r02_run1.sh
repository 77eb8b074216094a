#!/bin/bash
# round-2 GPU session 1: tests after the ADVICE fixes, baseline bench, skeleton structure experiments
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_fused.json 2> $O/bench_fused.err; echo "bench rc=$?" >> $O/rc.txt
timeout 300 tools/bin/hbm_mix > $O/hbm_mix.txt 2>&1; echo "mix rc=$?" >> $O/rc.txt
timeout 300 python tools/sweep.py --frames 1024 --rounds 4 --fpb 0,8,16,64 --rows 32 > $O/sweep_1024.txt 2>&1
timeout 300 python tools/sweep.py --frames 4096 --rounds 3 --iters 4 --fpb 0,32,64 --rows 32 > $O/sweep_4096.txt 2>&1
tail -3 $O/pytest.log; cat $O/rc.txt; cat $O/bench_fused.json; cat $O/hbm_mix.txt; cat $O/sweep_1024.txt $O/sweep_4096.txt
