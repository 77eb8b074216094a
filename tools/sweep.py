#!/usr/bin/env python3
"""Within-process A/B of kernel variants / tuning knobs on the bench workload.

Variants are interleaved over several rounds in ONE process (cross-process noise
looks like a kernel property otherwise); reports median and min ms per launch and
the implied algorithmic GB/s.  Usage:
  python tools/sweep.py --frames 1024 --rounds 5 --fpb 0,16,32,64 --rows 32,60 --order 0,1 --kernel tiled,gather
"""
import argparse
import itertools
import os
import sys
import tempfile

if "--lib" in sys.argv:  # experiment build of libmdc_hip (mono_dataset_code_amd/build.py:build_variant)
    os.environ["MDC_LIB_HIP"] = sys.argv[sys.argv.index("--lib") + 1]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mono_dataset_code_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--fpb", default="0")
    ap.add_argument("--kernel", default="tiled")
    ap.add_argument("--lib", default="")
    ap.add_argument("--rows", default="32")
    ap.add_argument("--order", default="0", help="tile placement: 0 bands, 1 whole rows per XCD, 2 identity")
    ap.add_argument("--workload", default="fused", choices=["fused", "unmap"])
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_sweep_"))
    sys.stdout.flush()
    so = os.dup(1)
    os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
    import ctypes
    ctypes.CDLL(None).fflush(None)
    os.dup2(so, 1)
    ctx = capi.Context(0)
    ctx.bind(fov, photo)
    info = ctx.info()
    B, npi = a.frames, 1280 * 1024
    npo = 640 * 480 if a.workload == "fused" else npi
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    s = st.cuda_stream
    d_in = torch.empty(B * npi, dtype=torch.uint8, device=dev)
    d_out = torch.empty(B * npo, dtype=torch.float32, device=dev)
    ctx.synth_frames(d_in.data_ptr(), 0, B, npi, synth.SEED, s)
    flags = 7 | (8 if a.workload == "fused" else 0)
    alg = (int(info.src_bbox_bytes) + npo * 4) if a.workload == "fused" else npi * 5
    kmap = {"tiled": capi.KERNEL_TILED, "gather": capi.KERNEL_GATHER, "auto": capi.KERNEL_AUTO}
    variants = list(itertools.product(a.kernel.split(","), [int(x) for x in a.fpb.split(",")], [int(x) for x in a.rows.split(",")],
                                      [int(x) for x in a.order.split(",")]))
    times = {v: [] for v in variants}
    for r in range(a.rounds + 1):
        for v in variants:
            ctx.set_option(capi.OPT_KERNEL, kmap[v[0]])
            ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, v[1])
            ctx.set_option(capi.OPT_TILE_ROWS, v[2])
            ctx.set_option(capi.OPT_TILE_ORDER, v[3])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), B, flags, s)
            e0.record()
            for _ in range(a.iters):
                ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), B, flags, s)
            e1.record()
            torch.cuda.synchronize()
            if r:
                times[v].append(e0.elapsed_time(e1) / a.iters)
    print("%-8s %5s %4s %3s %10s %10s %9s %7s" % ("kernel", "fpb", "rows", "ord", "median_ms", "min_ms", "GB/s", "frac8T"))
    for v in variants:
        med, mn = float(np.median(times[v])), float(np.min(times[v]))
        gbs = alg * B / (med * 1e-3) / 1e9
        print("%-8s %5d %4d %3d %10.4f %10.4f %9.1f %7.3f" % (v[0], v[1], v[2], v[3], med, mn, gbs, gbs / 8000), flush=True)


if __name__ == "__main__":
    main()
