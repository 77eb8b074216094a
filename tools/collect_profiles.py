#!/usr/bin/env python3
"""Copy the judged artefacts of tools/profile_bench.sh runs from gpurun_out/ (scratch) into
profiles/ (tracked) and refresh profiles/hbm_traffic.json, which bench.py reads for
roofline.traffic.  usage: tools/collect_profiles.py <round-tag, e.g. r01b>"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
WORKLOADS = {"fused": "remap_tiled_kernel", "unmap": "unmap_xpose_kernel", "pyramid": "remap_strip_kernel", "seq50k": "remap_tiled_kernel"}
traffic_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
for wl, kname in WORKLOADS.items():
    d = os.path.join(ROOT, "gpurun_out", "profile_%s_%s" % (tag, wl))
    if not os.path.isdir(d):
        continue
    shutil.copy(os.path.join(d, "summary.json"), os.path.join(ROOT, "profiles", "%s_%s_summary.json" % (tag, wl)))
    for f in glob.glob(d + "/stats/**/*kernel_stats.csv", recursive=True):
        shutil.copy(f, os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, wl)))
    s = json.load(open(os.path.join(d, "summary.json")))
    # frames per launch: what bench.py said under the profiler (its JSON line is kept next to the stats)
    frames = code_id = None
    try:
        line = [l for l in open(os.path.join(d, "bench_under_profiler.json")) if l.startswith("{")][-1]
        frames = json.loads(line)["roofline"]["frames_per_launch"]
        code_id = json.loads(line).get("code_id")  # mdc_code_id() of the library that ran under the profiler: bench.py refuses the entry for any other build
    except (OSError, IndexError, KeyError, ValueError):
        pass
    if not frames or wl == "seq50k":
        # seq50k: the run also holds the tuner's 4096-frame launches of the same kernel, so a mean per launch means nothing per
        # frame; bench.py quotes the per-frame figure measured on the headline's launches for it
        continue
    # a step made of several launches (the strip path in prefetched chunks): every launch of the step counts, per frame
    lps = None
    try:
        lps = json.loads(line)["roofline"].get("launches_per_step")
    except (NameError, KeyError, ValueError):
        pass
    if lps:
        main = [k for k in s["kernels"] if k.startswith(kname) and "hbm_bytes_per_launch" in s["kernels"][k]]
        pre = [k for k in s["kernels"] if k.startswith("prefetch_rows_kernel") and "hbm_bytes_per_launch" in s["kernels"][k]]
        if main:
            k = main[0]
            v = s["kernels"][k]
            nsteps = v["calls"] / lps[kname]
            rd = v["hbm_read_bytes_per_launch"] * v["calls"]
            wr = v["hbm_write_bytes_per_launch"] * v["calls"]
            pre_rd = sum(s["kernels"][p]["hbm_read_bytes_per_launch"] * s["kernels"][p]["calls"] for p in pre)
            traffic["%s:%s" % (wl, k)] = {
                "bytes_per_frame": (rd + wr + pre_rd) / nsteps / frames,
                "read_bytes_per_frame": (rd + pre_rd) / nsteps / frames, "write_bytes_per_frame": wr / nsteps / frames,
                "of_which_prefetch_read_bytes_per_frame": pre_rd / nsteps / frames,
                "kernel": k, "code_id": code_id, "avg_us_under_profiler": v["avg_us"], "frames_per_launch": frames, "launches_per_step": lps,
                "source": "profiles/%s_%s_summary.json" % (tag, wl),
                "correction": "FETCH_SIZE KiB x1024 x2 (gfx950 half-count, verified profiles/r01_fetch_calibration.txt) + WRITE_SIZE KiB x1024; all "
                              "launches of a step (chunked remap + prefetch) summed",
            }
        continue
    for k, v in s["kernels"].items():
        if k.startswith(kname) and "hbm_bytes_per_launch" in v:
            # one entry per kernel INSTANTIATION: bench.py only quotes a figure measured on the instantiation it launched
            traffic["%s:%s" % (wl, k)] = {
                "bytes_per_frame": v["hbm_bytes_per_launch"] / frames,
                "read_bytes_per_frame": v["hbm_read_bytes_per_launch"] / frames,
                "write_bytes_per_frame": v["hbm_write_bytes_per_launch"] / frames,
                "kernel": k, "code_id": code_id, "avg_us_under_profiler": v["avg_us"], "frames_per_launch": frames,
                "source": "profiles/%s_%s_summary.json" % (tag, wl),
                "correction": "FETCH_SIZE KiB x1024 x2 (gfx950 half-count, verified profiles/r01_fetch_calibration.txt) + WRITE_SIZE KiB x1024",
            }
json.dump(traffic, open(traffic_path, "w"), indent=1)
print(json.dumps(traffic, indent=1))
