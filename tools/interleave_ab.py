#!/usr/bin/env python3
"""Frame assignment of the tiled kernel on buffers from the product's allocator: a workgroup takes a RUN of consecutive frames (the default,
with the tapered tail) or every G-th frame (MDC_OPT_FRAME_INTERLEAVE).  One process, one pair of buffers, steady clocks: blocks of ~0.5 s per
setting, alternating, HIP events around every launch.   python tools/interleave_ab.py [frames] [rounds] [shape ...]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from mono_dataset_code_amd import capi, synth  # noqa: E402

FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
SHAPES = sys.argv[3:] or ["128x32:96", "128x32:128", "128x16:128"]
NPI, NPO = 1280 * 1024, 640 * 480
d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_il_"))
with bench.quiet_stdout():
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
ctx = capi.Context(0)
ctx.bind(fov, photo)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
s = st.cuda_stream
flags = 15
b = ctx.alloc_placed(FRAMES, flags, capi.PLACE_AUTO, s)
print("buffers:", b.describe()["how"])
ctx.synth_frames(b.d_in, 0, FRAMES, NPI, synth.SEED, s)
alg = (int(ctx.info().src_bbox_bytes) + NPO * 4) * FRAMES
res = {}
for rnd in range(ROUNDS + 1):
    for shp in SHAPES:
        wh, fpb = shp.split(":")
        cols, rows = (int(x) for x in wh.split("x"))
        for il in (0, 1):
            ctx.set_option(capi.OPT_TILE_COLS, cols)
            ctx.set_option(capi.OPT_TILE_ROWS, rows)
            ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, int(fpb))
            ctx.set_option(capi.OPT_FRAME_INTERLEAVE, il)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.35:  # back to steady clocks after the re-plan's idle gap
                for _ in range(20):
                    ctx.process_batch(b.d_in, b.d_out, FRAMES, flags, s)
                torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
            for x, y in ev:
                x.record()
                ctx.process_batch(b.d_in, b.d_out, FRAMES, flags, s)
                y.record()
            torch.cuda.synchronize()
            if rnd:
                res.setdefault((shp, il), []).append(float(np.median([x.elapsed_time(y) for x, y in ev])))
for (shp, il), v in res.items():
    m = float(np.median(v))
    print("tile %-10s frames per workgroup %-4s %-12s: %.4f ms  (rounds: %s)  frac %.4f" % (shp.split(":")[0], shp.split(":")[1], "interleaved" if il else "runs + taper", m,
                                                                                          " ".join("%.4f" % x for x in v), alg / (m * 1e-3) / 8e12))
