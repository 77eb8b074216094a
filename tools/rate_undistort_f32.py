#!/usr/bin/env python3
"""Rate of UndistorterFOV::undistort<float> on device-resident float frames (the reference's two-pass
composition, mdc_undistort_batch_device_f32): 1280x1024 f32 -> 640x480 f32."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mono_dataset_code_amd import capi, synth  # noqa: E402

d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_rate_"))
so = os.dup(1)
os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
import ctypes  # noqa: E402
ctypes.CDLL(None).fflush(None)
os.dup2(so, 1)
ctx = capi.Context(0)
ctx.bind(fov, photo)
info = ctx.info()
B, npi, npo = 256, 1280 * 1024, 640 * 480
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
s = st.cuda_stream
d_raw = torch.empty(B * npi, dtype=torch.uint8, device="cuda")
d_f = torch.empty(B * npi, dtype=torch.float32, device="cuda")
d_out = torch.empty(B * npo, dtype=torch.float32, device="cuda")
ctx.synth_frames(d_raw.data_ptr(), 0, B, npi, synth.SEED, s)
ctx.unmap_batch(d_raw.data_ptr(), d_f.data_ptr(), B, 7, s)
alg = int(info.src_bbox_bytes) * 4 + npo * 4
for label, kernel, cols, rows, nbuf in (("gather", capi.KERNEL_GATHER, 64, 32, 0), ("tiled auto", capi.KERNEL_AUTO, 0, 0, 0),
                                        ("tiled 64x32", capi.KERNEL_AUTO, 64, 32, 0), ("tiled 64x16", capi.KERNEL_AUTO, 64, 16, 0),
                                        ("tiled 64x64", capi.KERNEL_AUTO, 64, 64, 0), ("tiled 128x16", capi.KERNEL_AUTO, 128, 16, 0),
                                        ("tiled 128x32", capi.KERNEL_AUTO, 128, 32, 0), ("tiled 64x32 b3", capi.KERNEL_AUTO, 64, 32, 3),
                                        ("tiled 64x16 b2", capi.KERNEL_AUTO, 64, 16, 2), ("tiled 128x16 b3", capi.KERNEL_AUTO, 128, 16, 3)):
    ctx.set_option(capi.OPT_KERNEL, kernel)
    ctx.set_option(capi.OPT_TILE_COLS, cols)
    ctx.set_option(capi.OPT_TILE_ROWS, rows)
    ctx.set_option(capi.OPT_WINDOW_BUFFERS, nbuf)
    i = ctx.info()
    label += " [%dx%d %s]" % (i.f32_tile_w, i.f32_tile_h, "tiled" if i.f32_tiled else "gather")
    for _ in range(100):
        ctx.undistort_batch_f32(d_f.data_ptr(), d_out.data_ptr(), B, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ctx.undistort_batch_f32(d_f.data_ptr(), d_out.data_ptr(), B, s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print("undistort<float> %-32s: %.4f ms / %d frames, algorithmic %.2f MB/frame (bbox x 4 B + output) -> %.0f GB/s = %.3f of 8 TB/s"
          % (label, ms, B, alg / 1e6, alg * B / ms / 1e6, alg * B / ms / 1e6 / 8000))
