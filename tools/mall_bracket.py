#!/usr/bin/env python3
"""What do the fused kernel's reads cost when they come from HBM, from the Infinity Cache, from the L2?

The SAME launch (4096 frames -> same workgroups, same addresses per workgroup, same instruction stream, same number of
fabric requests) over four inputs that differ only in how much PHYSICAL memory lies behind the 5.4 GB of frame addresses:
  hbm      4096 distinct frames (5.4 GB): every source line comes from DRAM once per launch
  mall     one 160-MiB chunk of 128 frames mapped 32 times (libmdc_bench: mdcb_alias_alloc): below the 256-MiB Infinity
           Cache -> all frame reads are served on-die, none by DRAM
  mall64   64 frames (80 MiB) x 64
  l2       8 frames (10 MiB) x 512: each XCD's band of the 8 frames (~1 MiB) stays in its 4-MiB L2 -> reads never cross the
           fabric (the write side alone)
Outputs are 4096 distinct frames every time.  Variants are interleaved over rounds in one process (boxes differ by +-5 %).
usage: python tools/mall_bracket.py [--frames 4096] [--rounds 5] [--shapes 128x16,320x16,640x8] [--fpb 0]
Under rocprofv3 --pmc (tools/gpu_session.sh ea) the per-dispatch counters tell the variants apart by their order.
"""
import argparse
import ctypes as C
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mono_dataset_code_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--shapes", default="128x16")
    ap.add_argument("--fpb", type=int, default=0)
    ap.add_argument("--variants", default="hbm,mall,mall64,l2")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_mall_"))
    so = os.dup(1)
    sys.stdout.flush()
    os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
    C.CDLL(None).fflush(None)
    os.dup2(so, 1)
    ctx = capi.Context(0)
    ctx.bind(fov, photo)
    B, npi, npo = a.frames, 1280 * 1024, 640 * 480
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    s = st.cuda_stream
    L = capi.bench_lib()
    inputs = {}
    keep = []
    for name in a.variants.split(","):
        if name == "hbm":
            t = torch.empty(B * npi, dtype=torch.uint8, device=dev)
            ctx.synth_frames(t.data_ptr(), 0, B, npi, synth.SEED, s)
            keep.append(t)
            inputs[name] = (t.data_ptr(), B)
            continue
        phys = {"mall": 128, "mall64": 64, "l2": 8}[name]
        rep = B // phys
        p, g = C.c_void_p(), C.c_int64()
        rc = L.mdcb_alias_alloc(0, phys * npi, rep, C.byref(p), C.byref(g))
        if rc != 0:
            print("# %s: mdcb_alias_alloc failed (%d, granularity %d): variant skipped" % (name, rc, g.value))
            continue
        ctx.synth_frames(p.value, 0, phys, npi, synth.SEED, s)  # the first mapping; the others alias it
        torch.cuda.synchronize()
        inputs[name] = (p.value, phys)
    d_out = torch.empty(B * npo, dtype=torch.float32, device=dev)
    info = ctx.info()
    alg = int(info.src_bbox_bytes) + npo * 4
    flags = 15
    shapes = [tuple(int(v) for v in x.split("x")) for x in a.shapes.split(",")]
    # aliasing is only a change of physical backing: the outputs of frame f must be those of frame f % phys
    if "hbm" in inputs and "mall" in inputs:
        ctx.process_batch(inputs["mall"][0], d_out.data_ptr(), 256, flags, s)
        torch.cuda.synchronize()
        x = d_out[:256 * npo].view(256, npo)
        same = torch.equal(x[:128].view(torch.int32), x[128:256].view(torch.int32))
        ctx.process_batch(inputs["hbm"][0], d_out.data_ptr(), 128, flags, s)
        torch.cuda.synchronize()
        same = same and torch.equal(d_out[:128 * npo].view(torch.int32), x[128:256].reshape(-1).view(torch.int32))
        print("# aliased input gives the frames' own results (bitwise): %s" % same)
    for _ in range(150):  # clocks
        ctx.process_batch(next(iter(inputs.values()))[0], d_out.data_ptr(), B, flags, s)
    torch.cuda.synchronize()
    times = {}
    for r in range(a.rounds + 1):
        for cols, rows in shapes:
            ctx.set_option(capi.OPT_TILE_COLS, cols)
            ctx.set_option(capi.OPT_TILE_ROWS, rows)
            ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, a.fpb)
            for name, (ptr, _) in inputs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ctx.process_batch(ptr, d_out.data_ptr(), B, flags, s)
                e0.record()
                for _ in range(a.iters):
                    ctx.process_batch(ptr, d_out.data_ptr(), B, flags, s)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times.setdefault((cols, rows, name), []).append(e0.elapsed_time(e1) / a.iters)
    print("%-9s %-8s %8s %10s %10s %8s  %s" % ("tile", "input", "phys_MB", "median_ms", "us/frame", "frac8T", "kernel"))
    for cols, rows in shapes:
        ctx.set_option(capi.OPT_TILE_COLS, cols)
        ctx.set_option(capi.OPT_TILE_ROWS, rows)
        kn = ctx.describe_launch(flags, 0)
        for name, (_, phys) in inputs.items():
            med = float(np.median(times[(cols, rows, name)]))
            print("%-9s %-8s %8.0f %10.4f %10.4f %8.3f  %s" % ("%dx%d" % (cols, rows), name, phys * npi / 1e6, med, med * 1e3 / B,
                                                             alg * B / (med * 1e-3) / 8e12, kn), flush=True)


if __name__ == "__main__":
    main()
