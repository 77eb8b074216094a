#!/usr/bin/env python3
"""DSO-style preprocessing rate (row f4): base + 3 box levels + (I, dx, dy) / absSquaredGrad of every level, 1280x1024 base.
  (a) as round 2 did it: mdc_process_pyramid_batch_device over the whole batch, then one mdc_gradients_batch_device per level
      (every level is read back from HBM);
  (b) mdc_process_pyramid_gradients_batch_device: chunks of frames whose levels stay in the Infinity Cache, one gradient launch
      per chunk for all levels.
usage: python tools/dso_rate.py [frames]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mono_dataset_code_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
so = os.dup(1)
os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_dso_"), synth.camera_lines(1280, 1024, 1280, 1024))
fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
import ctypes  # noqa: E402
ctypes.CDLL(None).fflush(None)
os.dup2(so, 1)
ctx = capi.Context(0)
ctx.bind(fov, photo)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
s = st.cuda_stream
npi = 1280 * 1024
dims = [(1280 >> l, 1024 >> l) for l in range(4)]
PLACED = bool(os.environ.get("DSO_PLACED"))  # every buffer from the product's allocator (striped over the memory classes)


class P:
    """a device buffer by its address, with the two tensor methods this script uses"""

    def __init__(self, ptr, count):
        self.ptr, self.count = ptr, count

    def data_ptr(self):
        return self.ptr

    def view(self, dt):
        return torch.from_numpy(ctx.copy_to_host(self.ptr, self.count, np.int32))


if PLACED:
    pb = ctx.alloc_placed(n, 15, capi.PLACE_AUTO, s)
    print("buffers:", pb.describe()["how"])
    counts = [n * w * h for w, h in dims[1:]] + [n * w * h * 3 for w, h in dims] + [n * w * h for w, h in dims]
    ss = ctx.alloc_striped_set([4 * x for x in counts], s)
    ss2 = ctx.alloc_striped_set([4 * x for x in counts[3:]], s)  # (a set holds 16 buffers at most)
    bufs = [P(ss.d_ptr[k], x) for k, x in enumerate(counts)] + [P(ss2.d_ptr[k], x) for k, x in enumerate(counts[3:])]
    d_in, d_base = P(pb.d_in, n * npi // 4), P(pb.d_out, n * npi)
    lv, dI, ab, dI2, ab2 = bufs[0:3], bufs[3:7], bufs[7:11], bufs[11:15], bufs[15:19]
else:
    d_in = torch.empty(n * npi, dtype=torch.uint8, device="cuda")
    d_base = torch.empty(n * npi, dtype=torch.float32, device="cuda")
    lv = [torch.empty(n * w * h, dtype=torch.float32, device="cuda") for w, h in dims[1:]]
    dI = [torch.empty(n * w * h * 3, dtype=torch.float32, device="cuda") for w, h in dims]
    ab = [torch.empty(n * w * h, dtype=torch.float32, device="cuda") for w, h in dims]
    dI2 = [torch.empty_like(t) for t in dI]
    ab2 = [torch.empty_like(t) for t in ab]
ctx.synth_frames(d_in.data_ptr(), 0, n, npi, synth.SEED, s)


def separate():
    ctx.process_pyramid_batch(d_in.data_ptr(), d_base.data_ptr(), 4, [t.data_ptr() for t in lv], n, 15, s)
    for l, (w, h) in enumerate(dims):
        ctx.gradients_batch((d_base if l == 0 else lv[l - 1]).data_ptr(), w, h, dI[l].data_ptr(), ab[l].data_ptr(), n, s)


def fused(chunk=0):
    ctx.process_pyramid_gradients_batch(d_in.data_ptr(), d_base.data_ptr(), 4, [t.data_ptr() for t in lv], [t.data_ptr() for t in dI2],
                                        [t.data_ptr() for t in ab2], n, 15, chunk, s)


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


out_bytes = sum(w * h * 4 for w, h in dims) + sum(w * h * 16 for w, h in dims)
alg = 629990 + out_bytes
print("DSO preprocessing, %d frames of 1280x1024 -> base + 3 levels + gradient images (%.1f MB written per frame)" % (n, out_bytes / 1e6))
t = timeit(separate)
print("separate launches over the whole batch                  : %8.3f ms  %7.1f frames/s  %.2f TB/s algorithmic" % (t, n / t * 1e3, alg * n / t / 1e9))
timeit(lambda: fused(0), reps=3)  # (the one-call path's own output buffers and launch shapes: first touch, not timed)
for chunk in (0, 8, 16, 24, 48, 96, 0):
    t = timeit(lambda: fused(chunk))
    print("one call, chunks of %4s frames (two launches per chunk)  : %8.3f ms  %7.1f frames/s  %.2f TB/s algorithmic" % (chunk or "auto", t, n / t * 1e3, alg * n / t / 1e9))
torch.cuda.synchronize()
same = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(dI + ab, dI2 + ab2))
print("one-call results == separate-launch results, bit for bit:", same)
sys.exit(0 if same else 1)
