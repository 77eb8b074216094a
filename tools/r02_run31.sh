#!/bin/bash
# indexed vignette step: parity + rate
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r31
timeout 600 python -m pytest tests/test_vcal.py -x -q -m gpu > gpurun_out/r31/pytest.txt 2>&1
tail -15 gpurun_out/r31/pytest.txt
timeout 900 python tools/vcal_rate.py 200 0 2>&1 | grep -v "residual terms" > gpurun_out/r31/vcal_rate.txt
cat gpurun_out/r31/vcal_rate.txt
