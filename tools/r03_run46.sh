#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3_46; mkdir -p $O
python - <<'P' > $O/dir.txt 2>/dev/null
import os, sys
sys.argv = ["x", "256"]
os.environ["MDC_RATE_KINDS"] = ""
src = open("tools/reader_rate.py").read().split("for kind in os.environ.get")[0]
exec(src.replace("os.path.abspath(__file__)", "os.path.abspath(\"tools/reader_rate.py\")"))
d, avg = make("zip_jpg")
print(d)
P
D=$(tail -1 $O/dir.txt)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $GRAFT_REPO_ROOT/oracle/_ref/reader_rate_fast $D 1111 3 batch > $O/run.txt 2>&1
cat $O/run.txt | grep READER_RATE | head -2
python3 - <<P
import csv, glob
for f in glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us avg", round(float(r["TotalDurationNs"])/1e6,2), "ms total")
P
