#!/bin/bash
# Where a wave's frame loop spends its cycles: builds the MDC_EXP_TIMING diagnosis variant (s_memtime stamps around the
# phases of tile_frames, printed by wave 0 and wave 5 of some workgroups) and averages the printed lines.
# usage (on the GPU box): bash tools/phase_timing.sh        (the variant must have been built at home: see below)
# at home first:  python -c "from mono_dataset_code_amd import build; build.build_variant('timing', ['MDC_EXP_TIMING=1'])"
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/phase_timing; mkdir -p $O
export MDC_LIB_HIP=$PWD/mono_dataset_code_amd/variants/libmdc_hip_timing.so
timeout 300 python tools/launch_target.py fused 3 2>&1 | grep TIMING > $O/fused.txt
timeout 300 python tools/launch_target.py pyramid 3 2>&1 | grep TIMING > $O/pyramid.txt
python - <<'PY'
import re
for name in ("fused", "pyramid"):
    rows = []
    for l in open("gpurun_out/phase_timing/%s.txt" % name):
        m = re.search(r"wave (\d+) frames (\d+) cycles/frame: issue (\d+) compute\+stores (\d+) vmwait (\d+) barrier (\d+) total (\d+)", l)
        if m:
            rows.append(tuple(int(x) for x in m.groups()))
    for w in (0, 5):
        r = [x for x in rows if x[0] == w]
        if r:
            n = len(r)
            print(name, "wave", w, "samples", n, "frames", r[0][1],
                  "mean cycles/frame: DMA issue (+ level 3) %.0f | taps, LUT, arithmetic, store issue %.0f | s_waitcnt vmcnt %.0f | s_barrier %.0f | total %.0f"
                  % tuple(sum(x[i] for x in r) / n for i in (2, 3, 4, 5, 6)))
PY
