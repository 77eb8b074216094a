#!/bin/bash
# L1 -> L2 -> fabric read traffic of the fused kernel and of the fused-pyramid kernel (1024 frames per launch):
# how much of what the L1s request does the L2 absorb?  Each pass is its own rocprofv3 run (--pmc + --kernel-trace only).
# usage (GPU box): bash tools/pmc_l2.sh
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_l2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in fused pyramid; do
i=0
for C in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_READ_sum TCC_WRITE_sum TCC_REQ_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${wl}_pass$i -- python $GRAFT_REPO_ROOT/tools/launch_target.py $wl 4 > $OUT/${wl}_pass$i.log 2>&1
done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
with open(out + "/summary.txt", "w") as w:
    for wl in ("fused", "pyramid"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(out + "/%s_pass*/**/*counter_collection.csv" % wl, recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "remap_tiled" not in k: continue
                agg["remap_tiled_kernel"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            w.write("%s  (%s, 1024 frames per launch)\n" % (k, wl))
            for c, v in sorted(cs.items()):
                w.write("  %-32s mean/dispatch %.6g   per frame %.6g  (n=%d)\n" % (c, sum(v) / len(v), sum(v) / len(v) / 1024, len(v)))
print(open(out + "/summary.txt").read())
PY
