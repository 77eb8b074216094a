#!/bin/bash
# Round profile of the bench command (run on the GPU box via gpurun):
#   1. rocprofv3 --kernel-trace --stats            -> per-kernel average duration
#   2. rocprofv3 --pmc FETCH_SIZE   (own pass)     -> HBM read bytes per launch
#   3. rocprofv3 --pmc WRITE_SIZE   (own pass)     -> HBM write bytes per launch
# and a summary JSON with the gfx950 corrections of MI355X_MICROARCH.md (section HBM), checked on
# this box with tools/fetch_calib.hip: FETCH_SIZE counts KiB and reports exactly 1/2 of the bytes
# read (all widths we use), WRITE_SIZE counts KiB and is exact.
# usage: tools/profile_bench.sh <tag> [bench.py args...]
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --preroll-s 0.3 --preroll-max-s 2 --no-cpu-baseline --no-ceiling --no-secondary --placement-candidates 1 $*"
# --placement-candidates 1: the profiled run takes the first allocations as they come.  bench.py's search over candidate buffer pairs
# (mdc_tune_placement_device: 36 pairs x 7 launches of the SAME kernel, most of them on slow pairs) would be averaged into rocprof's per-kernel
# statistics; the profiled run's own bench line (bench_under_profiler.json) is the one its average must agree with.
# The tuner's trial launches (3 shapes x 4 frames-per-workgroup settings) would be averaged into the per-kernel statistics: pick the
# plan first, unprofiled, then profile a run that is told that plan and launches nothing else.
PLAN=$($BENCH --steps 2 --warmup 1 --preroll-s 0.05 --preroll-max-s 0.05 2> /dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        p = json.loads(l)['config']['plan']
        if isinstance(p, dict): print('--tile-cols %d --tile-rows %d --fpb %d' % (p['tile'][0], p['tile'][1], p['frames_per_workgroup']))
")
BENCH="$BENCH $PLAN"
echo "profiled command: $BENCH" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_under_profiler.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $BENCH > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $BENCH > /dev/null 2> $OUT/write.log
python3 - "$OUT" "$TAG" "$*" <<'PY'
import csv, glob, json, sys, collections
out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
import re
def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+(<[^()]*>)?)", k)
    return (m.group(1) if m else k).replace("mdc::", "")
stats = {}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                   "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for which in ("fetch", "write"):
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % which, recursive=True):
        for r in csv.DictReader(open(f)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"tag": tag, "bench_args": args, "kernels": {}}
for k, s in stats.items():
    e = dict(s)
    if k in pmc:
        f = pmc[k].get("FETCH_SIZE", []); w = pmc[k].get("WRITE_SIZE", [])
        # skip the warmup-free first dispatches? all launches are identical batches: plain mean
        if f: e["FETCH_SIZE_KiB_raw_mean"] = sum(f) / len(f); e["hbm_read_bytes_per_launch"] = 2.0 * 1024 * sum(f) / len(f)
        if w: e["WRITE_SIZE_KiB_raw_mean"] = sum(w) / len(w); e["hbm_write_bytes_per_launch"] = 1024.0 * sum(w) / len(w)
        if f and w: e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
    res["kernels"][k] = e
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
