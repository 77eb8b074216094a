"""The tracked rocprofv3 evidence of the CURRENT round (profiles/<tag>_*: tools/profile_round.py + tools/collect_profiles.py) is one
consistent set: made by ONE build -- the build of this tree --, its per-kernel times agree with the HIP events of the same run, the
roofline fraction recomputed from the csv is the bench line's, and every file profiles/INDEX.md cites exists.  (VERDICT r05, item 1:
five stale files and a kernel time longer than the driver's whole step.)  No GPU needed: this reads committed files."""
import csv
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
TAG = "r06"
WORKLOADS = ["fused", "unmap", "undistort_f32", "pyramid", "dso", "seq50k"]


pytestmark = pytest.mark.skipif(not glob.glob(os.path.join(PROF, TAG + "_*_summary.json")) and not os.environ.get("MDC_REQUIRE_ROUND_PROFILES"),
                                reason="no %s profiles recorded yet (tools/profile_round.py + tools/collect_profiles.py)" % TAG)


def _summaries():
    return {os.path.basename(f)[len(TAG) + 1:-len("_summary.json")]: json.load(open(f)) for f in sorted(glob.glob(os.path.join(PROF, TAG + "_*_summary.json")))}


def _bench_line(wl):
    return json.loads([l for l in open(os.path.join(PROF, "%s_%s_bench_under_profiler.json" % (TAG, wl))) if l.startswith("{")][-1])


def test_round_profiles_exist_for_every_workload():
    s = _summaries()
    assert set(WORKLOADS) <= set(s), sorted(s)  # (+ e.g. fused_128x16: the headline on the plan the tuner did not pick that day)
    for wl in WORKLOADS:
        for suffix in ("kernel_stats.csv", "kernel_stats_whole_run.csv", "bench_under_profiler.json", "summary.json", "profiled_command.txt"):
            assert os.path.exists(os.path.join(PROF, "%s_%s_%s" % (TAG, wl, suffix))), (wl, suffix)


def test_one_build_made_them_and_it_is_this_tree():
    from mono_dataset_code_amd import build

    here = build.code_id()
    ids = {}
    for wl, s in _summaries().items():
        ids["%s summary" % wl] = s["code_id"]
        ids["%s bench line" % wl] = _bench_line(wl)["code_id"]
        assert s["build_flags"] == "" and _bench_line(wl)["build_flags"] == "", wl  # the product build, no experiment switches
    traffic = json.load(open(os.path.join(PROF, "hbm_traffic.json")))
    mine = {k: v for k, v in traffic.items() if v.get("source", "").startswith("profiles/%s_" % TAG)}
    assert len(mine) >= len(WORKLOADS) - 1, sorted(mine)
    for k, v in mine.items():
        ids["hbm_traffic %s" % k] = v["code_id"]
    assert set(ids.values()) == {here}, {k: v for k, v in ids.items() if v != here}


@pytest.mark.parametrize("wl", WORKLOADS)
def test_rocprof_agrees_with_the_hip_events_of_the_same_run(wl):
    s = _summaries()[wl]
    line = _bench_line(wl)
    rf = line["roofline"]
    assert s["bench_line"]["kernel_ms"] == rf["kernel_ms"] and s["bench_line"]["frac"] == rf["frac"]  # the summary quotes THIS line
    # the csv (timed region only) -> the step's time: one launch per step = the kernel's average; several overlapping launches = summary's span
    rows = list(csv.DictReader(open(os.path.join(PROF, "%s_%s_kernel_stats.csv" % (TAG, wl)))))
    assert rows and all("timed steps" in r["Scope"] for r in rows)
    steps = s["steps_timed"]
    if len(rows) == 1 and int(rows[0]["Calls"]) == steps:
        step_us = float(rows[0]["AverageNs"]) / 1e3
        assert abs(step_us - s["rocprof_step_us"]) < 1e-3 * step_us
    else:
        step_us = s["rocprof_step_us"]  # (the span of the timed region / steps: the launches of a step overlap on two streams)
        busy = sum(float(r["TotalDurationNs"]) for r in rows) / 1e3 / steps
        assert 0.9 * step_us <= busy <= 2.2 * step_us, (busy, step_us)  # the kernels' own time: at least the step, at most two streams' worth of it
    tol = 0.02 if wl in ("fused", "unmap", "undistort_f32", "seq50k") else 0.04  # (chunked two-stream steps: the span includes launch gaps)
    assert abs(step_us - rf["kernel_ms"] * 1e3) <= tol * rf["kernel_ms"] * 1e3, (wl, step_us, rf["kernel_ms"])
    frac = rf["algorithmic_bytes_per_frame"] * rf["frames_per_launch"] / (step_us * 1e-6) / 8e12
    assert abs(frac - rf["frac"]) <= 0.01 + (0.01 if tol > 0.02 else 0), (wl, frac, rf["frac"])
    par = line["parity"]
    assert isinstance(par, dict) and par["mismatching_pixels"] == 0, par


def test_counter_traffic_is_sane():
    """fabric bytes per frame from the PMC passes: at least the algorithmic bytes (nothing is skipped), writes within 1 % of the algorithmic writes"""
    for wl, s in _summaries().items():
        hb = s["hbm_bytes_per_frame"]
        if wl == "seq50k" and not hb:  # (the FETCH_SIZE pass returns no row for the one 65-GB launch; bench.py quotes the fused entry -- the same
            continue                    # instantiation, bytes per FRAME -- for it, as rounds 3-5 did)
        assert hb, wl
        line = _bench_line(wl)["roofline"]
        alg, alg_r = line["algorithmic_bytes_per_frame"], line["algorithmic_read_bytes_per_frame"]
        assert hb["total"] >= 0.98 * alg, (wl, hb, alg)
        if wl != "dso":  # (the DSO step's gradient launch re-reads the levels: reads above the algorithmic figure are its known cost)
            assert abs(hb["write"] - (alg - alg_r)) <= 0.01 * (alg - alg_r), (wl, hb, alg - alg_r)
        assert hb["total"] < 2.0 * alg, (wl, hb, alg)


def test_index_cites_only_files_that_exist():
    txt = open(os.path.join(PROF, "INDEX.md")).read()
    cited = set(re.findall(r"`((?:profiles/)?[\w./-]+\.(?:json|csv|txt|patch|md))`", txt))
    missing = []
    for c in sorted(cited):
        p = c if c.startswith("profiles/") else "profiles/" + c
        if "*" in p or "<" in p:
            continue
        if not (os.path.exists(os.path.join(ROOT, p)) or os.path.exists(os.path.join(ROOT, c))):
            missing.append(c)
    assert not missing, missing
