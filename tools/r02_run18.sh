#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run18; mkdir -p $O
timeout 1200 python -m pytest tests/test_native_multi.py tests/test_reader.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
D=$(python -c "import tempfile;from mono_dataset_code_amd import synth;print(synth.write_sequence_calibration(tempfile.mkdtemp()))" 2>/dev/null | tail -1)
timeout 600 oracle/_ref/multi_gpu_seq $D 50000 5 > $O/multi_50k.txt 2>&1
timeout 600 python bench.py --workload seq50k --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_seq50k.json 2> $O/bench_seq50k.err
grep -n "passed\|failed\|Error\|assert" $O/pytest.log | head -20; cat $O/rc.txt; grep MULTI_GPU $O/multi_50k.txt; tail -3 $O/multi_50k.txt; cat $O/bench_seq50k.json; tail -3 $O/bench_seq50k.err
