#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run19; mkdir -p $O
D=$(python -c "import tempfile;from mono_dataset_code_amd import synth;print(synth.write_sequence_calibration(tempfile.mkdtemp()))" 2>/dev/null | tail -1)
timeout 600 oracle/_ref/multi_gpu_seq $D 50000 5 > $O/multi_50k.txt 2>&1
timeout 600 python bench.py --workload seq50k --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_seq50k.json 2> $O/bench_seq50k.err
timeout 600 python bench.py --steps 10 --warmup 2 --frames 16384 --no-cpu-baseline > $O/bench_16k.json 2> /dev/null
grep MULTI_GPU $O/multi_50k.txt; cat $O/bench_seq50k.json $O/bench_16k.json | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['config']['frames_per_gpu_per_step'], d['value'], r['frac'], r['kernel_ms'], r.get('frac_of_same_box_mix_ceiling'))"
