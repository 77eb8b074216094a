#!/usr/bin/env python3
"""Experiment: the fused kernel reading the raw frame from / writing the result to PAGE-LOCKED HOST memory directly (the
device-batch entry point given host pointers: mdc_host_alloc memory is mapped into the device's address space), against
mdc_process_host's copy in -> kernel -> copy out.  Per call: n frames, synchronous."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mono_dataset_code_amd import capi, synth  # noqa: E402

d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_zc_"))
fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
ctx = capi.Context(0)
ctx.bind(fov, photo)
W, H, w, h = fov.dims()
M = 64
frames = synth.noise_frames(0, 8, W * H)
pin_in = capi.PinnedArray((M, W * H), np.uint8)
pin_out = capi.PinnedArray((M, w * h), np.float32)
ref_out = capi.PinnedArray((M, w * h), np.float32)
for i in range(M):
    pin_in.array[i] = frames[i % 8]
st = torch.cuda.Stream()
s = st.cuda_stream
in_ptr = pin_in.array.ctypes.data
out_ptr = pin_out.array.ctypes.data
d_in = torch.empty(M * W * H, dtype=torch.uint8, device="cuda")
d_out = torch.empty(M * w * h, dtype=torch.float32, device="cuda")

for i in range(M):
    ctx.process_host(pin_in.array[i], ref_out.array[i], 15)


def bench(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def same(a, b):
    return bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))


print("product calls with MDC_OPT_ZERO_COPY on / off (page-locked buffers), 1280x1024 u8 -> 640x480 f32")
for zc in (2, 0):
    ctx.set_option(capi.OPT_ZERO_COPY, zc)
    for n in (1, 16, 64, 256):
        ins = [pin_in.array[i % M] for i in range(n)]
        outs = [pin_out.array[i % M] for i in range(n)] if n <= M else None
        if n == 1:
            t = bench(lambda: ctx.process_host(pin_in.array[0], pin_out.array[0], 15), 300)
            print("  zero copy %s: mdc_process_host                     %8.1f us per frame (%7.0f frames/s)" % ("off" if zc else "on ", t * 1e6, 1 / t))
        elif outs is not None:
            t = bench(lambda: ctx.process_frames_host(ins, outs, 15), max(10, 400 // n))
            print("  zero copy %s: mdc_process_frames_host, %3d frames   %8.1f us per frame (%7.0f frames/s)  same bits: %s" %
                  ("off" if zc else "on ", n, t / n * 1e6, n / t, same(pin_out.array[:n], ref_out.array[:n])))
ctx.set_option(capi.OPT_ZERO_COPY, 0)
print("fused 1280x1024 u8 -> 640x480 f32, page-locked host buffers, synchronous calls of n frames")
for n in (1, 2, 4, 16, 64):
    def copy_path():
        for i in range(n):
            ctx.process_host(pin_in.array[i], pin_out.array[i], 15)

    def zero_copy():
        ctx.process_batch(in_ptr, out_ptr, n, 15, s)
        st.synchronize()

    def zero_copy_in():  # reads from host memory, result stays on the device
        ctx.process_batch(in_ptr, d_out.data_ptr(), n, 15, s)
        st.synchronize()

    def zero_copy_out():  # frame already on the device, result written to host memory
        ctx.process_batch(d_in.data_ptr(), out_ptr, n, 15, s)
        st.synchronize()

    def frames_host():
        ctx.process_frames_host([pin_in.array[i] for i in range(n)], [pin_out.array[i] for i in range(n)], 15)

    reps = max(20, 400 // n)
    t_copy = bench(copy_path, reps)
    pin_out.array[:n] = 0
    t_zc = bench(zero_copy, reps)
    ok = same(pin_out.array[:n], ref_out.array[:n])
    t_in = bench(zero_copy_in, reps)
    t_out = bench(zero_copy_out, reps)
    t_fh = bench(frames_host, reps)
    print("n %3d: mdc_process_host x n %8.1f us (%7.0f frames/s) | process_frames_host %8.1f us (%7.0f) | kernel on host pointers %8.1f us (%7.0f frames/s, "
          "same bits: %s) | host in only %8.1f us | host out only %8.1f us" % (n, t_copy * 1e6, n / t_copy, t_fh * 1e6, n / t_fh, t_zc * 1e6, n / t_zc, ok,
                                                                             t_in * 1e6, t_out * 1e6), flush=True)
