#!/usr/bin/env python3
"""Device Huffman decoder alone: time per call of mdc_jpeg_huffman_batch_device on n streams of 1280x1024 frames (the
textured frames of tools/reader_rate.py, quality 90, ~265 KB each as grayscale) -- one component, with restart intervals,
YCbCr -- and the host decoder's time per frame for comparison."""
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

from mono_dataset_code_amd import capi  # noqa: E402


def textured(seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:1024, 0:1280]
    return np.clip(127 + 100 * np.sin(0.01 * x + seed) * np.cos(0.013 * y) + rng.normal(0, 4, (1024, 1280)), 0, 255).astype(np.uint8)


w, h = 1280, 1024
rec_bytes, pitch, rows = capi.jpeg_record_bytes(w, h)
ctx = capi.Context(0)
st = torch.cuda.current_stream().cuda_stream
N = 256
REPS = int(os.environ.get("REPS", "5"))


def run(label, save_kw, rgb=False, counts=(1, 4, 16, 32, 64, 128, 256)):
    if os.environ.get("COUNTS"):
        counts = tuple(int(v) for v in os.environ["COUNTS"].split(","))
    files = []
    for s in range(8):
        img = textured(s)
        if rgb:
            img = np.stack([img, np.roll(img, 5, 1), 255 - img], -1)
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=int(os.environ.get("Q", "90")), **save_kw)
        files.append(b.getvalue())
    cap = (2 * capi.JPEG_STREAM_HEADER_BYTES + 4 * 20480 + max(len(f) for f in files) + 64 + 15) & ~15
    streams = np.zeros((N, cap), np.uint8)
    one = np.zeros((8, cap), np.uint8)
    t0 = time.perf_counter()
    for i in range(8):
        capi.jpeg_stream(files[i], one[i])
    t_stream = (time.perf_counter() - t0) / 8
    for i in range(N):
        streams[i] = one[i % 8]
    rec = np.zeros(rec_bytes, np.uint8)
    t0 = time.perf_counter()
    for i in range(8):
        capi.decode_jpeg_record(files[i], rec, pitch)
    t_host = (time.perf_counter() - t0) / 8
    print("== %s: %d KB per file; host: mdch_jpeg_stream %.3f ms, mdch_decode_jpeg_record (Huffman decoding) %.3f ms per frame (Python call included)" %
          (label, sum(len(f) for f in files) / 8 / 1000, t_stream * 1e3, t_host * 1e3))
    d_streams = torch.from_numpy(streams).cuda()
    d_rec = torch.empty((N, rec_bytes), dtype=torch.uint8, device="cuda")
    d_status = torch.zeros((N,), dtype=torch.int32, device="cuda")
    d_frames = torch.empty((N, w * h), dtype=torch.uint8, device="cuda")
    for n in counts:
        for _ in range(2):
            ctx.jpeg_huffman_batch(d_streams.data_ptr(), cap, d_rec.data_ptr(), rec_bytes, w, h, pitch, rows, n, d_status.data_ptr(), st)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(REPS):
            ctx.jpeg_huffman_batch(d_streams.data_ptr(), cap, d_rec.data_ptr(), rec_bytes, w, h, pitch, rows, n, d_status.data_ptr(), st)
        e1.record()
        for _ in range(5):
            ctx.jpeg_idct_batch(d_rec.data_ptr(), rec_bytes, d_frames.data_ptr(), w, h, pitch, rows, n, st)
        e2.record()
        torch.cuda.synchronize()
        th, ti = e0.elapsed_time(e1) / REPS, e1.elapsed_time(e2) / 5
        print("n %3d: Huffman launches %8.3f ms (%6.1f us per frame, %7.0f frames/s) | inverse DCT %7.3f ms | status ok: %s" %
              (n, th, th / n * 1e3, n / th * 1e3, ti, bool(((d_status[:n] & 255) == 0).all())), flush=True)
    r = (d_status[:8] >> 8).cpu().numpy().tolist()
    if any(r):
        print("relaxation rounds of the 8 files (experiment build):", r)


# (the device-pointer entry point cannot look into the streams: it launches the kernels of all three kinds of stream; the
# reader's pipeline launches only the ones a chunk needs)
if os.environ.get("HUFF_KINDS") != "color":
    run("one component (what the TUM mono dataset ships)", {})
if os.environ.get("HUFF_KINDS") == "gray":
    sys.exit(0)
if os.environ.get("HUFF_KINDS") == "color":
    run("YCbCr 4:2:0, interleaved", {"subsampling": 2}, rgb=True, counts=(16, 64, 256))
    run("YCbCr 4:4:4, interleaved", {"subsampling": 0}, rgb=True, counts=(16, 64, 256))
    sys.exit(0)
run("one component, a restart interval per MCU row", {"restart_marker_rows": 1}, counts=(16, 64, 256))
run("YCbCr 4:2:0, interleaved", {"subsampling": 2}, rgb=True, counts=(16, 64, 256))
run("YCbCr 4:2:0, a restart interval per MCU row", {"subsampling": 2, "restart_marker_rows": 1}, rgb=True, counts=(64,))
