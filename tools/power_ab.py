#!/usr/bin/env python3
"""Power / clocks / launch time per library build of the headline launch (is the launch held back by the 1400-W
board limit -- i.e. does a loop that spends less energy per frame run the memory side faster -- or is it indifferent to sclk?).

  python tools/power_ab.py --libs default,mono_dataset_code_amd/variants/libmdc_hip_lutrep16.so,... [--frames 4096] [--rounds 3] [--seconds 1.2]

One process, one pair of buffers (from the product's allocator), one plan (tile shape / frames per workgroup given or the built-in one) for
every build.  Per round and build: launches queued back to back for ~`seconds`; while they run, the host reads the card's sysfs twice
(power1_average, pp_dpm_sclk / mclk / fclk starred levels: bench.gpu_clock_snapshot) -- the clocks UNDER this build's load; launch
time = HIP events around every launch of the second half.  Builds are interleaved over the rounds.  Diagnosis builds (fake compute, a stream
compiled out) compute wrong results on purpose: they are here for their energy, not for their output.
"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from mono_dataset_code_amd import capi, synth  # noqa: E402
from sweep import binding, try_set  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="default")
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=1.2)
    ap.add_argument("--shape", default="128x32")
    ap.add_argument("--fpb", type=int, default=96)
    ap.add_argument("--placement", default="auto")
    a = ap.parse_args()
    libs = a.libs.split(",")
    dev = torch.device("cuda", 0)
    d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_power_"))
    with bench.quiet_stdout():
        fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
        photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
    blob = capi.pack_tables(fov, photo)
    cols, rows = (int(x) for x in a.shape.split("x"))
    ctxs = {}
    for l in libs:
        m = binding(l)
        c = m.Context(0)
        c.import_tables(blob)
        try_set(m, c, "OPT_TILE_COLS", cols)
        try_set(m, c, "OPT_TILE_ROWS", rows)
        c.set_option(m.OPT_FRAMES_PER_BLOCK, a.fpb)
        ctxs[l] = (m, c)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    s = st.cuda_stream
    B, npi, npo = a.frames, 1280 * 1024, 640 * 480
    flags = capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED | capi.RECTIFY
    c0 = ctxs[libs[0]][1] if libs[0] in ("default", "-", "") else None
    own = capi.Context(0)
    own.import_tables(blob)
    which = {"first": capi.PLACE_FIRST, "malloc": capi.PLACE_MALLOC, "vmm": capi.PLACE_VMM, "auto": capi.PLACE_AUTO}[a.placement]
    pb = own.alloc_placed(B, flags, which, s)
    own.synth_frames(pb.d_in, 0, B, npi, synth.SEED, s)
    print("buffers:", pb.describe()["how"])
    alg = (int(own.info().src_bbox_bytes) + npo * 4) * B
    idle = bench.gpu_clock_snapshot(0)
    print("idle:", bench.format_clock_snapshot(idle))
    rec = {l: [] for l in libs}
    for r in range(a.rounds + 1):
        for l in libs:
            m, ctx = ctxs[l]
            ctx.process_batch(pb.d_in, pb.d_out, B, flags, s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ctx.process_batch(pb.d_in, pb.d_out, B, flags, s)
            e1.record()
            torch.cuda.synchronize()
            est = e0.elapsed_time(e1) / 8
            n = max(16, int(a.seconds * 1e3 / est))
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            t0 = time.perf_counter()
            for x, y in ev:
                x.record()
                ctx.process_batch(pb.d_in, pb.d_out, B, flags, s)
                y.record()
            snaps = []
            for frac in (0.45, 0.75):  # the launches are queued; the host reads the card while they run
                while time.perf_counter() - t0 < frac * a.seconds:
                    time.sleep(0.005)
                snaps.append(bench.gpu_clock_snapshot(0))
            torch.cuda.synchronize()
            ms = np.array([x.elapsed_time(y) for x, y in ev])[n // 2:]
            if r:
                rec[l].append((float(np.median(ms)), snaps))
    print("%-30s %10s %8s %8s | %s" % ("build", "median_ms", "frac8T", "flags", "power W / sclk / mclk / fclk per round (two samples each)"))
    for l in libs:
        m, ctx = ctxs[l]
        med = float(np.median([x[0] for x in rec[l]]))
        cells = []
        for _, snaps in rec[l]:
            cells.append(" ".join("%s/%s/%s/%s" % (sn.get("power_w"), sn.get("sclk_mhz"), sn.get("mclk_mhz"), sn.get("fclk_mhz")) for sn in snaps))
        try:
            bf = m.build_flags()
        except Exception:
            bf = "?"
        print("%-30s %10.4f %8.4f %8s | %s   [%s]" % (os.path.basename(l)[-30:], med, alg / (med * 1e-3) / 8e12, "", " ; ".join(cells), bf), flush=True)
    own.free_placed(pb)


if __name__ == "__main__":
    main()
