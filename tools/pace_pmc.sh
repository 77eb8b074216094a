#!/bin/bash
# fabric read requests of the headline launch per pacing setting (rocprofv3 --pmc TCC_EA0_RDREQ_sum, own pass per setting)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pace_pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in ${PACE_CFGS:-0:2:12 4:2:12 4:4:12 2:4:12}; do
  IFS=: read -r every sleep maxlead <<< "$cfg"
  d=$OUT/pmc_${every}_${sleep}_${maxlead}
  MDC_TILE_SYNC=$every MDC_TILE_SYNC_SLEEP=$sleep MDC_TILE_SYNC_MAXLEAD=$maxlead timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $d -- python $GRAFT_REPO_ROOT/tools/placed_probe.py first 1 ${PACE_FRAMES:-4096} > $d.log 2>&1
  python3 - $d "$cfg" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "remap_tiled" in r["Kernel_Name"] and r["Counter_Name"].startswith("TCC_EA0_RDREQ")]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
v = [float(r["Counter_Value"]) for r in rows[-30:]]
print("pacing %-8s: fabric read requests per launch (last 30 launches): mean %.4g  min %.4g max %.4g  = %.3f MB per frame" % (sys.argv[2], sum(v) / len(v), min(v), max(v), sum(v) / len(v) * 128 / 4096 / 1e6))
PY
done
