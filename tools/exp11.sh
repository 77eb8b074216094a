#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_bench.sh r01b_fused > gpurun_out/prof_fused.log 2>&1
bash tools/profile_bench.sh r01b_unmap --workload unmap --frames 512 > gpurun_out/prof_unmap.log 2>&1
bash tools/profile_bench.sh r01b_pyramid --workload pyramid --frames 256 > gpurun_out/prof_pyr.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_fused.json 2> gpurun_out/bench_fused.err
python bench.py --workload unmap --frames 512 --no-cpu-baseline > gpurun_out/bench_unmap.json 2> gpurun_out/bench_unmap.err
python bench.py --workload pyramid --frames 256 --no-cpu-baseline > gpurun_out/bench_pyramid.json 2> gpurun_out/bench_pyramid.err
cat gpurun_out/bench_*.json
tail -3 gpurun_out/bench_fused.err
