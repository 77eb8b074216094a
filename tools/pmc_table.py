#!/usr/bin/env python3
"""Per-dispatch counter table from rocprofv3 --pmc passes: tools/pmc_table.py <dir> <pass-prefix> [kernel-substring]
Rows are the dispatches of the matching kernel in launch order (one launch per variant of tools/mall_bracket.py --iters 1
--rounds 1 is: warm-up launches, then per shape and input the untimed + timed pair); columns are the counters of all passes."""
import collections
import csv
import glob
import sys

out, prefix = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else "remap_tiled"
cols = collections.OrderedDict()
names = {}
for f in sorted(glob.glob("%s/%s*/**/*counter_collection.csv" % (out, prefix), recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if want not in r["Kernel_Name"]:
            continue
        per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), r["Kernel_Name"]))
    for c, v in per.items():
        v.sort()
        cols[c] = [x[1] for x in v]
        names[c] = [x[2] for x in v]
if not cols:
    print("no counters found under", out)
    sys.exit(0)
n = min(len(v) for v in cols.values())
print("dispatch  " + "  ".join("%22s" % c for c in cols))
first = next(iter(names.values()))
for i in range(n):
    k = first[i]
    k = k[k.find("remap_"):][:60] if "remap_" in k else k[:60]
    print("%8d  " % i + "  ".join("%22.6g" % cols[c][i] for c in cols) + "   " + k)
