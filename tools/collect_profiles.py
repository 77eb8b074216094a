#!/usr/bin/env python3
"""Copy the judged artefacts of a tools/profile_round.py session from gpurun_out/<tag>_profiles/ (scratch) into profiles/ (tracked) and refresh
profiles/hbm_traffic.json, which bench.py reads for roofline.traffic (per kernel instantiation AND build: entries carry mdc_code_id()).
usage: tools/collect_profiles.py <round-tag, e.g. r06>"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag + "_profiles")
traffic_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
for f in sorted(glob.glob(os.path.join(src, tag + "_*"))):
    shutil.copy(f, os.path.join(ROOT, "profiles", os.path.basename(f)))
for f in sorted(glob.glob(os.path.join(src, tag + "_*_summary.json"))):
    s = json.load(open(f))
    hb = s.get("hbm_bytes_per_frame")
    if not hb:
        continue
    wl, kname = s["workload"], s["bench_line"]["kernel"]
    # one entry per workload and kernel INSTANTIATION (as bench.py names it): bench.py only quotes a figure measured on the instantiation
    # AND the build it launched; every launch of a step (chunked remap + prefetch, gradient launches) is in the sum
    traffic["%s:%s" % (wl, kname)] = {
        "bytes_per_frame": hb["total"], "read_bytes_per_frame": hb["read"], "write_bytes_per_frame": hb["write"],
        "kernel": kname, "code_id": s.get("code_id"), "step_us_under_profiler": s["rocprof_step_us"], "frames_per_launch": s["bench_line"]["frames_per_launch"],
        "per_kernel": {k: {x: v[x] for x in ("calls_per_step", "avg_us", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch") if x in v} for k, v in s["kernels"].items()},
        "source": "profiles/%s" % os.path.basename(f), "correction": s["corrections"],
    }
json.dump(traffic, open(traffic_path, "w"), indent=1)
print("\n".join("%s: %.0f B/frame (%s)" % (k, v["bytes_per_frame"], v.get("code_id")) for k, v in traffic.items() if v.get("source", "").startswith("profiles/" + tag)))
