#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_39; mkdir -p $O
python - <<'P' > $O/trace.txt 2>&1
import os, sys, subprocess
sys.argv = ["x", "256"]
os.environ["MDC_RATE_KINDS"] = ""
sys.path.insert(0, "tools")
import importlib.util
src = open("tools/reader_rate.py").read().split("for kind in os.environ.get")[0]
exec(src.replace("os.path.abspath(__file__)", "os.path.abspath(\"tools/reader_rate.py\")"))
d, avg = make("zip_jpg")
for stage in ("0", "1", "2"):
    for threads in ("0",):
        r = subprocess.run([os.path.join(BIN, "reader_rate_fast"), d, "1111", "3", "batch"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           env=dict(os.environ, MDC_GPU_JPEG=stage, MDC_READER_TRACE="1"))
        print("== stage", stage)
        print("\n".join(l for l in r.stdout.splitlines() if "getImages" in l or "READER_RATE" in l))
P
cat $O/trace.txt | grep -v amdgpu
