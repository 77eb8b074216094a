#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp13; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | head -20 > $O/pytest.txt
python bench.py --workload pyramid --frames 256 --no-cpu-baseline > $O/bench_pyramid.json 2> $O/bench_pyramid.err
python bench.py --no-cpu-baseline > $O/bench_fused.json 2> /dev/null
cat $O/pytest.txt $O/bench_pyramid.json $O/bench_fused.json; tail -3 $O/bench_pyramid.err
