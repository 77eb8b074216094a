#!/usr/bin/env python3
"""Time per frame of the device-batch kernels as a function of the frames per launch (= memory footprint), to see
where cross-launch cache residency (256 MiB Infinity Cache: a bench step re-reads the same input) stops flattering
the number and what the steady state is.
usage: python tools/footprint_curve.py [frames,frames,...]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mono_dataset_code_amd import capi, synth  # noqa: E402

counts = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [128, 256, 384, 512, 1024]
so = os.dup(1)
os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
d1 = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_fc_"))
d2 = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_fc2_"), synth.camera_lines(1280, 1024, 1280, 1024))
objs = []
ctxs = {}
for key, d in (("640", d1), ("1280", d2)):
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
    c = capi.Context(0)
    c.bind(fov, photo)
    objs += [fov, photo]
    ctxs[key] = c
import ctypes  # noqa: E402
ctypes.CDLL(None).fflush(None)
os.dup2(so, 1)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
s = st.cuda_stream
npi = 1280 * 1024
NMAX = max(counts)
d_in = torch.empty(NMAX * npi, dtype=torch.uint8, device="cuda")
ctxs["640"].synth_frames(d_in.data_ptr(), 0, NMAX, npi, synth.SEED, s)
d_big = torch.empty(NMAX * npi, dtype=torch.float32, device="cuda")
d_lv = [torch.empty(NMAX * (1280 >> l) * (1024 >> l), dtype=torch.float32, device="cuda") for l in (1, 2, 3)]
lv = [t.data_ptr() for t in d_lv]
bbox640 = int(ctxs["640"].info().src_bbox_bytes)
bbox1280 = int(ctxs["1280"].info().src_bbox_bytes)
ops = [
    ("unMapImage (1 B in, 4 B out per pixel)", npi * 5, lambda n: ctxs["640"].unmap_batch(d_in.data_ptr(), d_big.data_ptr(), n, 7, s)),
    ("fused -> 640x480", bbox640 + 640 * 480 * 4, lambda n: ctxs["640"].process_batch(d_in.data_ptr(), d_big.data_ptr(), n, 15, s)),
    ("fused -> 1280x1024 (no pyramid)", bbox1280 + npi * 4, lambda n: ctxs["1280"].process_batch(d_in.data_ptr(), d_big.data_ptr(), n, 15, s)),
    ("fused -> 1280x1024 + 3 levels", bbox1280 + npi * 4 + 1720320, lambda n: ctxs["1280"].process_pyramid_batch(d_in.data_ptr(), d_big.data_ptr(), 4, lv, n, 15, s)),
]
for _ in range(50):
    ops[0][2](256)
torch.cuda.synchronize()
print("%-42s %7s %10s %10s %8s" % ("kernel", "frames", "ms", "us/frame", "frac8T"))
for name, alg, fn in ops:
    for n in counts:
        fn(n)
        fn(n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(3, min(20, 4096 // n))
        e0.record()
        for _ in range(reps):
            fn(n)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%-42s %7d %10.4f %10.3f %8.3f" % (name, n, ms, ms * 1e3 / n, alg * n / (ms * 1e-3) / 8e12), flush=True)
