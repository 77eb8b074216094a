#!/usr/bin/env python3
"""General in-process A/B: any output geometry, builds and knobs, base-only or base + pyramid levels.
  python tools/exp.py --out 1280x1024 --frames 1024 --libs default,mono_dataset_code_amd/variants/libmdc_hip_skipstore.so \
      --two-stage 2,1 --cols 128 --rows 16 --nbuf 0 --fpb 0 --pyramid 0,1
Variants are interleaved over rounds in ONE process on ONE GPU; prints median / min ms per launch."""
import argparse
import itertools
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mono_dataset_code_amd import capi, synth  # noqa: E402
from sweep import binding, try_set  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="1280x1024")
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--libs", default="default")
ap.add_argument("--two-stage", default="0")
ap.add_argument("--cols", default="0")
ap.add_argument("--rows", default="0")
ap.add_argument("--nbuf", default="0")
ap.add_argument("--fpb", default="0")
ap.add_argument("--order", default="0")
ap.add_argument("--sched", default="0")
ap.add_argument("--prefetch", default="0", help="strip path: frames per prefetched chunk (0 auto, -1 off)")
ap.add_argument("--pfdist", default="-1", help="(experiment 09 builds only) strip kernel's own prefetch: frames ahead (0 auto, -1 off)")
ap.add_argument("--taper", default="0", help="tapered tail of large tiled launches (0 auto = on, 1 on, 2 off)")
ap.add_argument("--streams", default="0", help="streams the prefetched chunks alternate over (0 auto, 1, 2)")
ap.add_argument("--pyramid", default="0", help="0 = base only (mdc_process_batch_device), 1 = base + levels 1..3")
ap.add_argument("--flags", type=int, default=15)
a = ap.parse_args()
ow, oh = (int(x) for x in a.out.split("x"))
libs = a.libs.split(",")
so = os.dup(1)
os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_exp_"), synth.camera_lines(1280, 1024, ow, oh))
fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
import ctypes  # noqa: E402
ctypes.CDLL(None).fflush(None)
os.dup2(so, 1)
blob = capi.pack_tables(fov, photo)
ctxs = {}
for l in libs:
    m = binding(l)
    c = m.Context(0)
    c.import_tables(blob)
    ctxs[l] = (m, c)
B, npi, npo = a.frames, 1280 * 1024, ow * oh
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
s = st.cuda_stream
d_in = torch.empty(B * npi, dtype=torch.uint8, device="cuda")
d_out = torch.empty(B * npo, dtype=torch.float32, device="cuda")
lv = [torch.empty(B * (ow >> l) * (oh >> l), dtype=torch.float32, device="cuda") for l in (1, 2, 3)]
ctxs[libs[0]][1].synth_frames(d_in.data_ptr(), 0, B, npi, synth.SEED, s)
ints = lambda x: [int(v) for v in x.split(",")]  # noqa: E731
variants = list(itertools.product(libs, ints(a.two_stage), ints(a.cols), ints(a.rows), ints(a.nbuf), ints(a.fpb), ints(a.order), ints(a.sched), ints(a.pyramid), ints(a.prefetch), ints(a.pfdist), ints(a.streams), ints(a.taper)))
times = {v: [] for v in variants}
names = {}
ref_out, same = {}, {}


def launch(m, ctx, pyr):
    if pyr:
        ctx.process_pyramid_batch(d_in.data_ptr(), d_out.data_ptr(), 4, [t.data_ptr() for t in lv], B, a.flags, s)
    else:
        ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), B, a.flags, s)


m0, c0 = ctxs[libs[0]]
for _ in range(60):
    launch(m0, c0, 0)
torch.cuda.synchronize()
for r in range(a.rounds + 1):
    for v in variants:
        m, ctx = ctxs[v[0]]
        try_set(m, ctx, "OPT_TWO_STAGE", v[1])
        try_set(m, ctx, "OPT_TILE_COLS", v[2])
        try_set(m, ctx, "OPT_TILE_ROWS", v[3])
        try_set(m, ctx, "OPT_WINDOW_BUFFERS", v[4])
        ctx.set_option(m.OPT_FRAMES_PER_BLOCK, v[5])
        try_set(m, ctx, "OPT_TILE_ORDER", v[6])
        try_set(m, ctx, "OPT_FRAME_INTERLEAVE", v[7])
        try_set(m, ctx, "OPT_PREFETCH_CHUNK", v[9])
        try_set(m, ctx, "OPT_PREFETCH_DIST", v[10])
        try_set(m, ctx, "OPT_PREFETCH_STREAMS", v[11])
        try_set(m, ctx, "OPT_TAIL_TAPER", v[12])
        try:
            names[v] = ctx.describe_launch(a.flags, 4 if v[8] else 0)
        except Exception:  # noqa: BLE001
            names[v] = "?"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        launch(m, ctx, v[8])
        if r == 0:  # results of every variant against the first one's, bit for bit (NaN as a mask)
            torch.cuda.synchronize()
            cur = [d_out.view(torch.int32)] + ([t.view(torch.int32) for t in lv] if v[8] else [])
            if v[8] not in ref_out:
                ref_out[v[8]] = [t.clone() for t in cur]
            same[v] = all(bool(torch.equal(x, y)) for x, y in zip(cur, ref_out[v[8]]))
            d_out.zero_()
            for t in lv:
                t.zero_()
        e0.record()
        for _ in range(a.iters):
            launch(m, ctx, v[8])
        e1.record()
        torch.cuda.synchronize()
        if r:
            times[v].append(e0.elapsed_time(e1) / a.iters)
print("out %dx%d frames %d" % (ow, oh, B))
print("%-26s %2s %4s %4s %3s %3s %3s %3s %3s %4s %4s %3s %3s %9s %9s %4s  %s" % ("lib", "2s", "cols", "rows", "buf", "fpb", "ord", "sch", "pyr", "pref", "pfd", "str", "tap", "median_ms", "min_ms", "same", "kernel"))
for v in variants:
    print("%-26s %2d %4d %4d %3d %3d %3d %3d %3d %4d %4d %3d %3d %9.4f %9.4f %4s  %s" % (os.path.basename(v[0])[-26:], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12],
                                                                      float(np.median(times[v])), float(np.min(times[v])), "yes" if same.get(v) else "NO", names[v]), flush=True)
