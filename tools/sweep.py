#!/usr/bin/env python3
"""Within-process A/B of kernel builds / tuning knobs on the bench workload.

Variants are interleaved over several rounds in ONE process on ONE GPU (different boxes and
different processes differ by +-5 %, more than most effects of interest); reports median and
min ms per launch and the implied algorithmic GB/s.  Usage:
  python tools/sweep.py --frames 1024 --rounds 5 --fpb 0,16,32,64 --rows 32,60 --order 0,1 --kernel tiled,gather
  python tools/sweep.py --libs default,mono_dataset_code_amd/variants/libmdc_hip_v1.so   # builds side by side
Every lib is an independent ctypes binding of include/mdc_hip.h (same HIP runtime); the
calibration tables are built once on the host and imported into each as the broadcast blob.
"""
import argparse
import importlib.util
import itertools
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mono_dataset_code_amd import capi, synth  # noqa: E402


def binding(path):
    """A second, independent instance of the capi module bound to another libmdc_hip build."""
    if path in ("default", "-", ""):
        return capi
    spec = importlib.util.spec_from_file_location("capi_" + os.path.basename(path).replace(".", "_"),
                                                  os.path.join(ROOT, "mono_dataset_code_amd", "capi.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.LIB_HIP_PATH = path if os.path.isabs(path) else os.path.join(ROOT, path)
    return m


def try_set(m, ctx, name, value):
    opt = getattr(m, name, None)
    if opt is None:
        return
    try:
        ctx.set_option(opt, value)
    except Exception:  # older builds do not know newer options / values
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--settle", type=float, default=0.3, help="seconds of untimed launches of a variant before its timed ones (0: the old behaviour, short bursts)")
    ap.add_argument("--fpb", default="0")
    ap.add_argument("--kernel", default="tiled")
    ap.add_argument("--libs", default="default", help="comma list of libmdc_hip builds ('default' = the in-tree one)")
    ap.add_argument("--lib", default="", help="(compat) single alternative build")
    ap.add_argument("--rows", default="32")
    ap.add_argument("--cols", default="64", help="output tile columns: 64 or 128")
    ap.add_argument("--shapes", default="", help="tile shapes as COLSxROWS,... (overrides the --cols x --rows product), e.g. 128x16,320x16,640x8")
    ap.add_argument("--pad", default="0", help="LDS row pitch padded to 128 bytes: 0 / 1")
    ap.add_argument("--order", default="0", help="tile placement: 0 bands, 1 whole rows per XCD, 2 identity")
    ap.add_argument("--sched", default="-1", help="frame assignment: -1 library default, 0 consecutive runs, 1 interleaved")
    ap.add_argument("--nbuf", default="0", help="LDS window buffers (0 = automatic)")
    ap.add_argument("--remap", default="fov", choices=["fov", "affine", "affine128"],
                    help="diagnosis: replace the FOV remap by a distortion-free one of the same scale (no window overlap "
                         "from the bow); affine128 = output tile columns map to whole 128-byte source lines")
    ap.add_argument("--workload", default="fused", choices=["fused", "unmap"])
    ap.add_argument("--placement", default="", help="frame / result buffers from the product's allocator (first | malloc | vmm | auto) instead of torch.empty")
    a = ap.parse_args()
    libs = [a.lib] if a.lib else a.libs.split(",")
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
    blob = capi.pack_tables(fov, photo)
    ctxs = {}
    for l in libs:
        m = binding(l)
        c = m.Context(0)
        c.import_tables(blob)
        if a.remap != "fov":
            sx = 1.345 if a.remap == "affine" else 2.0  # 64 outputs -> 86 or exactly 128 source bytes
            x0 = 200.3 if a.remap == "affine" else 0.3
            xs = (x0 + sx * np.arange(640, dtype=np.float64)).astype(np.float32)
            ys = (146.9 + 1.515 * np.arange(480, dtype=np.float64)).astype(np.float32)
            rx = np.broadcast_to(xs[None, :], (480, 640)).copy()
            ry = np.broadcast_to(ys[:, None], (480, 640)).copy()
            c.set_remap(rx, ry, 1280, 1024, 640, 480)
        ctxs[l] = (m, c)
    info = ctxs[libs[0]][1].info()
    B, npi = a.frames, 1280 * 1024
    npo = 640 * 480 if a.workload == "fused" else npi
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    s = st.cuda_stream
    flags = 7 | (8 if a.workload == "fused" else 0)
    if a.placement:
        own = capi.Context(0)
        own.import_tables(blob)
        pb = own.alloc_placed(B, flags, {"first": capi.PLACE_FIRST, "malloc": capi.PLACE_MALLOC, "vmm": capi.PLACE_VMM, "auto": capi.PLACE_AUTO}[a.placement], s)
        print("buffers:", pb.describe()["how"], "probe first / chosen", pb.ms_first, pb.ms_chosen)

        class Ptr:  # the few tensor methods this script uses, on a raw device address
            def __init__(self, ptr, n, dtype):
                self.ptr, self.n, self.dtype = ptr, n, dtype

            def data_ptr(self):
                return self.ptr

            def host(self):
                return torch.from_numpy(own.copy_to_host(self.ptr, self.n, np.uint8 if self.dtype == torch.uint8 else np.float32))

            def clone(self):
                return self.host()

            def view(self, dt):
                return self.host().view(dt)

            def zero_(self):
                pass

        d_in, d_out = Ptr(pb.d_in, B * npi, torch.uint8), Ptr(pb.d_out, B * npo, torch.float32)
    else:
        d_in = torch.empty(B * npi, dtype=torch.uint8, device=dev)
        d_out = torch.empty(B * npo, dtype=torch.float32, device=dev)
    ctxs[libs[0]][1].synth_frames(d_in.data_ptr(), 0, B, npi, synth.SEED, s)
    alg = (int(info.src_bbox_bytes) + npo * 4) if a.workload == "fused" else npi * 5
    kmap = {"tiled": capi.KERNEL_TILED, "gather": capi.KERNEL_GATHER, "auto": capi.KERNEL_AUTO}
    ints = lambda x: [int(v) for v in x.split(",")]  # noqa: E731
    if a.shapes:
        shp = [tuple(int(v) for v in x.split("x")) for x in a.shapes.split(",")]
        variants = [(l, k, f, r, o, sc, nb, c, pd) for l, k, f, (c, r), o, sc, nb, pd in
                    itertools.product(libs, a.kernel.split(","), ints(a.fpb), shp, ints(a.order), ints(a.sched), ints(a.nbuf), ints(a.pad))]
    else:
        variants = list(itertools.product(libs, a.kernel.split(","), ints(a.fpb), ints(a.rows), ints(a.order), ints(a.sched), ints(a.nbuf), ints(a.cols), ints(a.pad)))
    times = {v: [] for v in variants}
    same, ref_out = {}, None
    # DVFS: the first ~30 ms after an idle period run ~15 % slow (profiles/r01_dvfs_warmup_curve.txt)
    m0, c0 = ctxs[libs[0]]
    for _ in range(200):
        c0.process_batch(d_in.data_ptr(), d_out.data_ptr(), B, flags, s)
    torch.cuda.synchronize()
    for r in range(a.rounds + 1):
        for v in variants:
            m, ctx = ctxs[v[0]]
            ctx.set_option(m.OPT_KERNEL, kmap[v[1]])
            ctx.set_option(m.OPT_FRAMES_PER_BLOCK, v[2])
            try_set(m, ctx, "OPT_TILE_COLS", 0)
            try_set(m, ctx, "OPT_TILE_ROWS", v[3])
            try_set(m, ctx, "OPT_TILE_COLS", v[7])
            try_set(m, ctx, "OPT_TILE_ORDER", v[4])
            if v[5] >= 0:
                try_set(m, ctx, "OPT_FRAME_INTERLEAVE", v[5])
            try_set(m, ctx, "OPT_WINDOW_BUFFERS", v[6])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), B, flags, s)
            if r and a.settle > 0:  # back to steady clocks after the re-plan's idle gap (the ~30 ms after one run 8 % slow: r06 experiment 06)
                import time
                t_s = time.perf_counter()
                while time.perf_counter() - t_s < a.settle:
                    for _ in range(10):
                        ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), B, flags, s)
                    torch.cuda.synchronize()
            if r == 0:  # every variant's output against the first variant's, bit for bit (tuning variants must not change results)
                torch.cuda.synchronize()
                if ref_out is None:
                    ref_out = d_out.clone()
                    same[v] = True
                else:
                    same[v] = bool(torch.equal(d_out.view(torch.int32), ref_out.view(torch.int32) if a.placement else ref_out.view(torch.int32)))
                d_out.zero_()
            e0.record()
            for _ in range(a.iters):
                ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), B, flags, s)
            e1.record()
            torch.cuda.synchronize()
            if r:
                times[v].append(e0.elapsed_time(e1) / a.iters)
    print("%-28s %-7s %5s %4s %4s %3s %3s %3s %3s %10s %10s %9s %7s" % ("lib", "kernel", "fpb", "cols", "rows", "ord", "sch", "buf", "pad", "median_ms", "min_ms", "GB/s", "frac8T") + "  ==first")
    for v in variants:
        med, mn = float(np.median(times[v])), float(np.min(times[v]))
        gbs = alg * B / (med * 1e-3) / 1e9
        print("%-28s %-7s %5d %4d %4d %3d %3d %3d %3d %10.4f %10.4f %9.1f %7.3f" % (os.path.basename(v[0])[-28:], v[1], v[2], v[7], v[3], v[4], v[5], v[6], v[8], med, mn, gbs, gbs / 8000) + "  " + str(same.get(v)), flush=True)
    if not all(same.values()):
        print("RESULTS DIFFER between variants")
        sys.exit(1)


if __name__ == "__main__":
    main()
