#!/bin/bash
# round-3 closing evidence, second take (after the two-stream strip path, gradient v2, zero copy, GPU Huffman stage):
# GPU tests, profiles of the three bench workloads (stats + PMC), secondary rates
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b_final; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
bash tools/final_profiles.sh r03b > $O/final.log 2>&1
timeout 600 python tools/vcal_rate.py 200 6 > $O/vcal_rate.txt 2>&1
timeout 600 python bench.py --workload seq50k --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r03b_seq50k.json 2> /dev/null
D=$(python -c "import tempfile;from mono_dataset_code_amd import synth;print(synth.write_sequence_calibration(tempfile.mkdtemp()))" 2>/dev/null | tail -1)
timeout 600 oracle/_ref/multi_gpu_seq $D 50000 5 2>&1 | grep MULTI_GPU > $O/multi_50k.txt
timeout 300 python tools/dso_rate.py 2>&1 | grep -v amdgpu.ids > $O/dso_rate.txt
timeout 900 python tools/reader_rate.py 256 2>&1 | grep -v amdgpu.ids > $O/reader_rate.txt
timeout 300 python tools/huffman_rate.py 2>&1 | grep -v amdgpu.ids > $O/huffman_rate.txt
timeout 300 python tools/zero_copy_rate.py 2>&1 | grep -v "amdgpu.ids\|^Input\|^Out\|resolution\|Reading\|Success" > $O/zero_copy_rate.txt
timeout 900 bash tools/soak_threads.sh 60 > $O/thread_soak.txt 2>&1
timeout 600 python tools/soak.py 24 > $O/soak.txt 2>&1
grep -n "passed\|failed" $O/pytest.log | tail -2; cat $O/rc.txt; cat $O/multi_50k.txt; tail -2 $O/thread_soak.txt; tail -1 $O/soak.txt
python -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_r03b_*.json')):
    d=json.loads(open(f).readline()); r=d['roofline']; print(f.split('/')[-1], r['frac'], r['kernel_ms'], r.get('frac_of_same_box_mix_ceiling'), r['kernel'], d.get('parity'))
"
