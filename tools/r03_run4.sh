#!/bin/bash
# round 3, run 4: phase timing of the direct kernel at the scale-1 geometry, with and without stores; two-stage parity again
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_04; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_stage" ) > $O/pytest.txt 2>&1; grep -E "passed|failed|Error|error" $O/pytest.txt | tail -3
for rep in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_stage" 2>&1 | grep -E "passed|failed" | tail -1; done
V=$PWD/mono_dataset_code_amd/variants
for var in timing timing_skipstore; do
for what in base1280 pyramid fused; do
MDC_LIB_HIP=$V/libmdc_hip_$var.so timeout 300 python tools/launch_target.py $what 3 2 2>&1 | grep TIMING > $O/${var}_$what.txt
python - <<PY
import re
rows=[]
for l in open("$O/${var}_$what.txt"):
    m = re.search(r"wave (\d+) frames (\d+) cycles/frame: issue (\d+) compute\+stores (\d+) vmwait (\d+) barrier (\d+) total (\d+)", l)
    if m: rows.append(tuple(int(x) for x in m.groups()))
for w in (0, 5):
    r=[x for x in rows if x[0]==w]
    if r:
        n=len(r)
        print("$var $what wave", w, "samples", n, "frames", r[0][1], "cycles/frame: issue %.0f | compute+store issue %.0f | vmwait %.0f | barrier %.0f | total %.0f" % tuple(sum(x[i] for x in r)/n for i in (2,3,4,5,6)))
PY
done; done
