#!/usr/bin/env python3
"""The round's rocprofv3 evidence for bench.py's workloads, made in ONE session of ONE build (run on the GPU box):

  python tools/profile_round.py <tag, e.g. r06> [workload ...]      default: fused unmap undistort_f32 pyramid dso seq50k

Per workload three runs of the same bench.py command (placement by the product's allocator, plan by the tuner: what the driver's line runs):
  1. rocprofv3 --kernel-trace --stats            per-dispatch times
  2. rocprofv3 --pmc FETCH_SIZE  --kernel-trace   fabric read requests per dispatch    (own pass, MI355X_MICROARCH.md's HBM section)
  3. rocprofv3 --pmc WRITE_SIZE  --kernel-trace   fabric write bytes per dispatch      (own pass)
bench.py --markers brackets its timed region with two launches of a no-op kernel (mdcb_marker_kernel): everything here is cut to the
dispatches BETWEEN the two -- the allocator's probe launches, the tuner's trials, pre-roll and warm-up (many of them on other buffers or
plans) do not enter the averages.  gfx950 corrections: FETCH_SIZE counts KiB and reports 1/2 of the bytes read (profiles/r01_fetch_calibration.txt),
WRITE_SIZE counts KiB and is exact.

Writes, under gpurun_out/<tag>_profiles/ (tools/collect_profiles.py <tag> copies them into profiles/ and refreshes profiles/hbm_traffic.json):
  <tag>_<wl>_kernel_stats.csv            per kernel, TIMED REGION ONLY (rocprofv3's column layout + Scope)
  <tag>_<wl>_kernel_stats_whole_run.csv  rocprofv3's own --stats table of the whole process
  <tag>_<wl>_bench_under_profiler.json   bench.py's line of run 1
  <tag>_<wl>_summary.json                code_id, command, per-kernel timed statistics and counters, per-step sums, agreement with the HIP events
  <tag>_<wl>_profiled_command.txt
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = {"fused": 40, "unmap": 40, "undistort_f32": 40, "pyramid": 30, "dso": 20, "seq50k": 12}


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+(<[^()]*>)?)", k)
    return (m.group(1) if m else k).replace("mdc::", "")


def find(d, pat):
    f = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return f[0] if f else None


def between_markers(rows, key_name="Kernel_Name"):
    """rows (dicts with Dispatch_Id) -> those dispatched between the first and the last mdcb_marker_kernel dispatch"""
    rows = sorted(rows, key=lambda r: int(r["Dispatch_Id"]))
    marks = [int(r["Dispatch_Id"]) for r in rows if "mdcb_marker_kernel" in r[key_name]]
    if len(marks) < 2:
        raise SystemExit("no marker pair in the trace (%d markers): was bench.py run with --markers?" % len(marks))
    lo, hi = marks[0], marks[-1]
    return [r for r in rows if lo < int(r["Dispatch_Id"]) < hi and "mdcb_marker_kernel" not in r[key_name]]


def run(cmd, out, err):
    with open(out, "w") as fo, open(err, "w") as fe:
        return subprocess.run(cmd, stdout=fo, stderr=fe, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp")).returncode


def main():
    tag = sys.argv[1]
    wls = sys.argv[2:] or ["fused", "unmap", "undistort_f32", "pyramid", "dso", "seq50k"]
    outdir = os.path.join(ROOT, "gpurun_out", tag + "_profiles")
    os.makedirs(outdir, exist_ok=True)
    for spec in wls:
        # "fused:128x16" = the workload on a given plan (tile columns x rows, 128 frames per workgroup) instead of the tuner's pick: the other
        # instantiation a device may choose gets its own traffic entry; files are named <tag>_<workload>_<shape>_*
        wl, _, shape = spec.partition(":")
        steps = STEPS.get(wl, 20)
        name = wl + ("_" + shape if shape else "")
        scratch = os.path.join(ROOT, "gpurun_out", "%s_profile_scratch_%s" % (tag, name))
        shutil.rmtree(scratch, ignore_errors=True)
        os.makedirs(scratch)
        bench = ["python3", os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", str(steps), "--warmup", "5", "--no-cpu-baseline", "--no-ceiling",
                 "--no-secondary", "--markers", "--parity-frames", "2"]
        if shape:
            bench += ["--tile-cols", shape.split("x")[0], "--tile-rows", shape.split("x")[1], "--fpb", "128"]
        base = os.path.join(outdir, "%s_%s" % (tag, name))
        open(base + "_profiled_command.txt", "w").write(" ".join(bench).replace(ROOT + "/", "") + "\n  under: rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE --kernel-trace | "
                                                       "--pmc WRITE_SIZE --kernel-trace  (three runs; statistics cut to the dispatches between bench.py's two marker launches)\n")
        rc = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", scratch + "/stats", "--"] + bench, base + "_bench_under_profiler.json", scratch + "/stats.log")
        rcs = [rc]
        for which in ("FETCH_SIZE", "WRITE_SIZE"):
            rcs.append(run(["rocprofv3", "--pmc", which, "--kernel-trace", "--output-format", "csv", "-d", scratch + "/" + which, "--"] + bench,
                           scratch + "/%s.out" % which, scratch + "/%s.log" % which))
        try:
            line = json.loads([l for l in open(base + "_bench_under_profiler.json") if l.startswith("{")][-1])
        except (IndexError, ValueError):
            print("%s: bench.py printed no line under the profiler (rc %s), see %s" % (wl, rcs, scratch))
            continue
        whole = find(scratch + "/stats", "*kernel_stats.csv")
        if whole:
            shutil.copy(whole, base + "_kernel_stats_whole_run.csv")
        trace = find(scratch + "/stats", "*kernel_trace.csv")
        rows = between_markers(list(csv.DictReader(open(trace))))
        per = collections.OrderedDict()
        for r in rows:
            per.setdefault(short(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        total = float(sum(sum(v) for v in per.values())) or 1.0
        scope = "dispatches between bench.py's two mdcb_marker_kernel launches = its %d timed steps" % steps
        with open(base + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_ALL)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "Scope"])
            for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([k, len(v), sum(v), "%.3f" % (sum(v) / len(v)), "%.2f" % (100.0 * sum(v) / total), min(v), max(v), scope])
        span_ns = max(int(r["End_Timestamp"]) for r in rows) - min(int(r["Start_Timestamp"]) for r in rows)
        counters = {}
        for which in ("FETCH_SIZE", "WRITE_SIZE"):
            cc = find(scratch + "/" + which, "*counter_collection.csv")
            if not cc:
                continue
            crow = list(csv.DictReader(open(cc)))
            tr = find(scratch + "/" + which, "*kernel_trace.csv")
            # markers carry no counters of interest but they are dispatches like any other: cut by their ids in the counter table itself
            ids = sorted({(int(r["Dispatch_Id"]), r["Kernel_Name"]) for r in crow})
            marks = [i for i, n in ids if "mdcb_marker_kernel" in n]
            if len(marks) < 2 and tr:
                marks = [int(r["Dispatch_Id"]) for r in csv.DictReader(open(tr)) if "mdcb_marker_kernel" in r["Kernel_Name"]]
            if len(marks) < 2:
                print("%s: no marker pair in the %s pass" % (wl, which))
                continue
            lo, hi = min(marks), max(marks)
            acc = collections.defaultdict(list)
            for r in crow:
                if r["Counter_Name"] == which and lo < int(r["Dispatch_Id"]) < hi:
                    acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            counters[which] = acc
        kernels = collections.OrderedDict()
        step_read = step_write = 0.0
        for k, v in per.items():
            e = {"calls_timed": len(v), "calls_per_step": len(v) / float(steps), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3,
                 "total_us_per_step": sum(v) / 1e3 / steps}
            f_, w_ = counters.get("FETCH_SIZE", {}).get(k), counters.get("WRITE_SIZE", {}).get(k)
            if f_:
                e["FETCH_SIZE_KiB_raw_mean"] = sum(f_) / len(f_)
                e["hbm_read_bytes_per_launch"] = 2.0 * 1024 * sum(f_) / len(f_)
                step_read += 2.0 * 1024 * sum(f_) / steps
            if w_:
                e["WRITE_SIZE_KiB_raw_mean"] = sum(w_) / len(w_)
                e["hbm_write_bytes_per_launch"] = 1024.0 * sum(w_) / len(w_)
                step_write += 1024.0 * sum(w_) / steps
            if f_ and w_:
                e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
            kernels[k] = e
        rf = line["roofline"]
        frames = rf["frames_per_launch"]
        single = len(per) == 1 and all(abs(e["calls_per_step"] - 1) < 1e-9 for e in kernels.values())
        # one launch per step: rocprof's average of that kernel IS the step; several launches (chunks over two streams): the span from the first
        # start to the last end of the region, per step (the launches overlap, their sum is not the step's time)
        step_us = list(kernels.values())[0]["avg_us"] if single else span_ns / 1e3 / steps
        summary = {
            "tag": tag, "workload": wl, "plan_given": shape or None, "code_id": line.get("code_id"), "build_flags": line.get("build_flags"), "steps_timed": steps,
            "scope": scope, "command": " ".join(bench).replace(ROOT + "/", ""),
            "bench_line": {"kernel": rf["kernel"], "kernel_ms": rf["kernel_ms"], "kernel_ms_median": rf["kernel_ms_median"], "frac": rf["frac"],
                           "value": line["value"], "ms_per_step": line["ms_per_step"], "frames_per_launch": frames,
                           "algorithmic_bytes_per_frame": rf["algorithmic_bytes_per_frame"], "placement": line["config"]["placement"], "plan": line["config"]["plan"],
                           "parity_mismatching_pixels": (line.get("parity") or {}).get("mismatching_pixels") if isinstance(line.get("parity"), dict) else line.get("parity")},
            "rocprof_step_us": step_us, "rocprof_step_is": "the kernel's average duration" if single else "span of the timed region / steps (launches of a step overlap on two streams)",
            "hip_events_step_us": rf["kernel_ms"] * 1e3, "rocprof_over_hip_events": step_us / (rf["kernel_ms"] * 1e3),
            "frac_from_rocprof": rf["algorithmic_bytes_per_frame"] * frames / (step_us * 1e-6) / 8e12,
            "hbm_bytes_per_frame": {"read": step_read / frames, "write": step_write / frames, "total": (step_read + step_write) / frames} if step_read and step_write else None,
            "corrections": "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count, profiles/r01_fetch_calibration.txt), WRITE_SIZE KiB x 1024; separate --pmc passes; every launch of a step summed",
            "kernels": kernels, "rocprofv3_return_codes": rcs,
        }
        json.dump(summary, open(base + "_summary.json", "w"), indent=1)
        print("%-14s %s: rocprof %.1f us / step vs HIP events %.1f us (%.3f), frac %.4f (line) %.4f (rocprof); fabric bytes / frame: %s; code_id %s" % (
            wl, rf["kernel"], step_us, rf["kernel_ms"] * 1e3, summary["rocprof_over_hip_events"], rf["frac"], summary["frac_from_rocprof"],
            summary["hbm_bytes_per_frame"], summary["code_id"]), flush=True)
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
