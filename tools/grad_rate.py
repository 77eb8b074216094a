#!/usr/bin/env python3
"""The DSO hand-off's gradient launch ALONE (mdc_gradients_batch_device / gradients_levels_kernel: per pixel 4 bytes read, 16 written)
per pyramid level, against a linear stream of the same bytes on the same buffers (libmdc_bench: mdcb_ceiling_mix_device).
  python tools/grad_rate.py [frames]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mono_dataset_code_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
ctx = capi.Context(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
s = st.cuda_stream


def timeit(fn, reps=8):
    fn()
    fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


print("gradient launch alone, %d frames per launch; traffic = 4 B read + 16 B written per pixel" % n)
for w, h in ((1280, 1024), (640, 512), (640, 480), (320, 256)):
    npx = w * h
    src = torch.rand(n * npx, dtype=torch.float32, device="cuda")
    dI = torch.empty(n * npx * 3, dtype=torch.float32, device="cuda")
    ab = torch.empty(n * npx, dtype=torch.float32, device="cuda")
    t = timeit(lambda: ctx.gradients_batch(src.data_ptr(), w, h, dI.data_ptr(), ab.data_ptr(), n, s))
    rb, wb = n * npx * 4, n * npx * 16
    best = None
    for blocks in (4096, 16384, 65536):
        for span in (0, 1):
            # the stream writes ONE buffer: dI holds 12 of the 16 bytes, so its 3/4 of the time scaled to all bytes
            c = timeit(lambda: ctx.ceiling_mix(src.data_ptr(), rb, dI.data_ptr(), n * npx * 12, blocks, span, s))
            c *= (rb + wb) / float(rb + n * npx * 12)
            best = c if best is None else min(best, c)
    print("%4d x %4d : kernel %.4f ms = %.2f TB/s of traffic (%.3f us per frame);  linear stream of the same bytes %.4f ms = %.2f TB/s;  kernel at %.3f of it"
          % (w, h, t, (rb + wb) / t / 1e9, t / n * 1e3, best, (rb + wb) / best / 1e9, best / t), flush=True)
    del src, dI, ab
    torch.cuda.empty_cache()
