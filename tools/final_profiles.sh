#!/bin/bash
# Round-end evidence: rocprofv3 stats + PMC passes of the three bench workloads, then plain bench lines.
# usage (on the GPU box): bash tools/final_profiles.sh <tag>     then, back home: python tools/collect_profiles.py <tag>
cd $GRAFT_REPO_ROOT
TAG=${1:-r01c}
bash tools/profile_bench.sh ${TAG}_fused > gpurun_out/prof_fused.log 2>&1
bash tools/profile_bench.sh ${TAG}_unmap --workload unmap --frames 1024 > gpurun_out/prof_unmap.log 2>&1
bash tools/profile_bench.sh ${TAG}_pyramid --workload pyramid > gpurun_out/prof_pyr.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_${TAG}_fused.json 2> /dev/null
python bench.py --workload unmap --frames 1024 --no-cpu-baseline > gpurun_out/bench_${TAG}_unmap.json 2> /dev/null
python bench.py --workload pyramid --no-cpu-baseline > gpurun_out/bench_${TAG}_pyramid.json 2> /dev/null
python tools/rate_undistort_f32.py 2>&1 | grep undistort > gpurun_out/rate_${TAG}_undistort_f32.txt
python tools/host_path_rate.py 2>&1 | grep "frames/s" > gpurun_out/rate_${TAG}_host_path.txt
cat gpurun_out/bench_${TAG}_*.json gpurun_out/rate_${TAG}_*.txt
