#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer drop-in calls (never the bench `value`):
single-frame, synchronous, pageable host buffers -- what a caller of the reference's
classes gets without changing a line."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from mono_dataset_code_amd import capi, synth  # noqa: E402

d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_host_"))
fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
ctx = capi.Context(0)
ctx.bind(fov, photo)
W, H, w, h = fov.dims()
frames = synth.noise_frames(0, 8, W * H)
tmp = np.zeros(W * H, np.float32)
out = np.zeros(w * h, np.float32)
N = 200


def rate(fn):
    fn(0)
    t0 = time.perf_counter()
    for i in range(N):
        fn(i)
    dt = time.perf_counter() - t0
    return N / dt, N * W * H / dt / 1e6


def two_calls(i):  # exactly what DatasetReader::getImage does with the drop-in classes
    photo.unmap(frames[i % 8], tmp, 1, 1, 1)
    fov.undistort(tmp, out)


M = 256  # frames per mdc_process_frames_host call
page_in = [frames[i % 8].copy() for i in range(M)]
page_out = [np.zeros(w * h, np.float32) for _ in range(M)]
pin_in = capi.PinnedArray((M, W * H), np.uint8)
pin_out = capi.PinnedArray((M, w * h), np.float32)
for i in range(M):
    pin_in.array[i] = frames[i % 8]


def many(raws, outs):
    ctx.process_frames_host(raws, outs, 15)  # warm
    t0 = time.perf_counter()
    for _ in range(4):
        ctx.process_frames_host(raws, outs, 15)
    dt = time.perf_counter() - t0
    return 4 * M / dt, 4 * M * W * H / dt / 1e6


# the same two class calls with the opt-in page-locking of the repeated W*H float buffer (what
# `MDC_PIN_CALLER_BUFFERS=1 playDataset ...` gives an unmodified reader)
os.environ["MDC_PIN_CALLER_BUFFERS"] = "1"
fov_p = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
photo_p = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
del os.environ["MDC_PIN_CALLER_BUFFERS"]
pin_out1 = capi.PinnedArray((w * h,), np.float32)  # a pooled ExposureImage::image is page-locked already


def two_calls_pinned(i):
    photo_p.unmap(frames[i % 8], tmp, 1, 1, 1)
    fov_p.undistort(tmp, pin_out1.array)


pin_in1 = capi.PinnedArray((8, W * H), np.uint8)
pin_in1.array[:] = frames


def fused_pinned(i):  # what this repo's DatasetReader::getImage does after decoding: page-locked frame -> pooled image
    ctx.process_host(pin_in1.array[i % 8], pin_out1.array, 15)


print("host path, 1280x1024 -> 640x480, g+v+o, %d frames each" % N, file=sys.stderr)
for name, fn in (("unMapImage + undistort<float> (two class calls, W*H float round trip)", two_calls),
                 ("  the same with MDC_PIN_CALLER_BUFFERS=1 and a pooled (page-locked) ExposureImage", two_calls_pinned),
                 ("mdc_process_host (fused, one call, pageable buffers)", lambda i: ctx.process_host(frames[i % 8], out, 15)),
                 ("mdc_process_host from / to page-locked buffers (the reader's getImage after decode)", fused_pinned)):
    fps, mpix = rate(fn)
    print("%-90s %8.1f frames/s  %9.1f Mpix/s" % (name, fps, mpix), file=sys.stderr)
for name, (raws, outs) in (("mdc_process_frames_host, %d frames per call, pageable buffers" % M, (page_in, page_out)),
                           ("mdc_process_frames_host, %d frames per call, mdc_host_alloc buffers" % M,
                            ([pin_in.array[i] for i in range(M)], [pin_out.array[i] for i in range(M)]))):
    fps, mpix = many(raws, outs)
    print("%-90s %8.1f frames/s  %9.1f Mpix/s" % (name, fps, mpix), file=sys.stderr)


# ---- several host threads on ONE context (SURVEY.md 8b "Threading"): every call leases its own stream + staging slot, so the
# uploads, kernels and downloads of different threads overlap (round 2 held one mutex across the whole call: no scaling)
import threading  # noqa: E402

for T in (1, 2, 4, 8, 12):
    ins = [capi.PinnedArray((W * H,), np.uint8) for _ in range(T)]
    outs = [capi.PinnedArray((w * h,), np.float32) for _ in range(T)]
    for k in range(T):
        ins[k].array[:] = frames[k % 8]
    per = 400
    go = threading.Barrier(T + 1)

    def work(k):
        ctx.process_host(ins[k].array, outs[k].array, 15)
        go.wait()
        for _ in range(per):
            ctx.process_host(ins[k].array, outs[k].array, 15)

    th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    for t in th:
        t.start()
    go.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print("%-90s %8.1f frames/s  %9.1f Mpix/s" % ("mdc_process_host, page-locked buffers, %2d host threads on one context" % T, T * per / dt, T * per * W * H / dt / 1e6),
          file=sys.stderr)
