#!/usr/bin/env python3
"""Where a getImages call on a zipped JPEG sequence spends its time: runs oracle/_ref/reader_rate_fast in batch mode with
MDC_READER_TRACE / MDC_PIPE_TRACE set and prints everything it says.  usage: python tools/reader_trace.py [frames] [passes] [batch|single]
(single: getImage frame by frame, no trace lines -- for steady-state rates over many passes; MDC_TRACE_ENV=K=V,K=V adds environment)"""
import os
import subprocess
import sys

sys.argv = [sys.argv[0]] + sys.argv[1:]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
PASSES = sys.argv[2] if len(sys.argv) > 2 else "4"
MODE = sys.argv[3] if len(sys.argv) > 3 else "batch"
os.environ["MDC_RATE_KINDS"] = ""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
src = open(os.path.join(ROOT, "tools", "reader_rate.py")).read().split("\nfor kind in os.environ.get")[0]
g = {"__name__": "reader_rate_lib", "__file__": os.path.join(ROOT, "tools", "reader_rate.py")}
sys.argv = [sys.argv[0], str(N)]
exec(compile(src, "reader_rate.py", "exec"), g)
d, avg = g["make"]("zip_jpg")
env = dict(os.environ, MDC_READER_TRACE="1", MDC_PIPE_TRACE="1") if MODE == "batch" else dict(os.environ)
env.update({k: v for k, v in (a.split("=", 1) for a in os.environ.get("MDC_TRACE_ENV", "").split(",") if "=" in a)})
r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "reader_rate_fast"), d, "1111", PASSES] + (["batch"] if MODE == "batch" else []), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                   text=True, timeout=900, env=env)
for l in r.stdout.splitlines():
    if "READER_RATE" in l or "DatasetReader::getImages" in l or "mdc_process" in l:
        print(l)
