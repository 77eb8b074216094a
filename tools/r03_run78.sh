#!/bin/bash
# rocprofv3 kernel stats of the reader's JPEG path: getImages over a zipped 1280x1024 JPEG sequence, 1024 frames per call, 6 passes
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3_78; mkdir -p $O
D=$(python - <<'PY'
import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.argv = ["x", "1024"]
os.environ["MDC_RATE_KINDS"] = ""
src = open(os.path.join(ROOT, "tools", "reader_rate.py")).read().split("\nfor kind in os.environ.get")[0]
g = {"__name__": "lib", "__file__": os.path.join(ROOT, "tools", "reader_rate.py")}
sys.path.insert(0, ROOT)
exec(compile(src, "reader_rate.py", "exec"), g)
print(g["make"]("zip_jpg")[0])
PY
)
D=$(echo "$D" | tail -1)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $GRAFT_REPO_ROOT/oracle/_ref/reader_rate_fast $D 1111 6 batch > $O/run.txt 2> $O/stats.log
grep -a READER_RATE $O/run.txt | head -2
F=$(find $O/stats -name "*kernel_stats.csv" | head -1)
cp "$F" $O/reader_jpeg_kernel_stats.csv
python3 - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-90s calls %6s  avg %10.1f us  total %8.2f ms  %5s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
