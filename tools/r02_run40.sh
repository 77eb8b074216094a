#!/bin/bash
# phase timing of the frame loop (s_memtime stamps in wave 0 and wave 5 of some workgroups)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r40; mkdir -p $O
export MDC_LIB_HIP=$PWD/mono_dataset_code_amd/variants/libmdc_hip_timing.so
timeout 300 python tools/pc_sample_target.py fused 3 2>&1 | grep TIMING | sort | uniq -c | sort -rn | head -400 > $O/fused_all.txt
timeout 300 python tools/pc_sample_target.py pyramid 3 2>&1 | grep TIMING > $O/pyr_all.txt
python - <<'PY'
import re,collections
for name in ("fused_all","pyr_all"):
    rows=[]
    for l in open("gpurun_out/r40/%s.txt"%name):
        m=re.search(r"wave (\d+) frames (\d+) cycles/frame: issue (\d+) compute\+stores (\d+) vmwait (\d+) barrier (\d+) total (\d+)",l)
        if m: rows.append(tuple(int(x) for x in m.groups()))
    for w in (0,5):
        r=[x for x in rows if x[0]==w]
        if not r: continue
        n=len(r)
        print(name,"wave",w,"samples",n,"frames",r[0][1],"mean cycles/frame: issue %.0f compute+stores %.0f vmwait %.0f barrier %.0f total %.0f"%tuple(sum(x[i] for x in r)/n for i in (2,3,4,5,6)))
PY
