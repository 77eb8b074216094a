#!/usr/bin/env python3
"""Config 5 (1280x1024 -> 1280x1024 + 3 box levels, mdc_process_pyramid_batch_device): frames per prefetched chunk x streams x frames per
workgroup, interleaved rounds on PAIRS of buffers that lie apart (a pair's place decides 8 % of a launch's time: profiles/r05_experiments/10_*).
  python tools/pyramid_sweep.py [frames [rounds]]        PYR_CHUNKS=0,24,36,48  PYR_STREAMS=2,1  PYR_FPB=0
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mono_dataset_code_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
CHUNKS = [int(x) for x in os.environ.get("PYR_CHUNKS", "0,20,28,36,44,52,64,96").split(",")]
STREAMS = [int(x) for x in os.environ.get("PYR_STREAMS", "2,1").split(",")]
FPBS = [int(x) for x in os.environ.get("PYR_FPB", "0").split(",")]
so = os.dup(1)
os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_pyr_"), synth.camera_lines(1280, 1024, 1280, 1024))
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
pairs = []


class P:  # a raw device address with the one tensor method this script uses
    def __init__(self, ptr):
        self.ptr = ptr

    def data_ptr(self):
        return self.ptr


if os.environ.get("PYR_PLACED"):  # every buffer from the product's allocator (mdc_alloc_placed_device + mdc_alloc_striped_set_device)
    pb = ctx.alloc_placed(n, 15, capi.PLACE_AUTO, s)
    ctx.synth_frames(pb.d_in, 0, n, npi, synth.SEED, s)
    ss = ctx.alloc_striped_set([n * w * h * 4 for w, h in dims[1:]], s)
    pairs.append((P(pb.d_in), P(pb.d_out), [P(ss.d_ptr[k]) for k in range(3)]))
    print("buffers:", pb.describe()["how"])
for k in range(0 if os.environ.get("PYR_PLACED") else 2):  # two (frames, base) pairs, a 40-GB spacer between them (given back)
    d_in = torch.empty(n * npi, dtype=torch.uint8, device="cuda")
    ctx.synth_frames(d_in.data_ptr(), 0, n, npi, synth.SEED, s)
    d_base = torch.empty(n * npi, dtype=torch.float32, device="cuda")
    lv = [torch.empty(n * w * h, dtype=torch.float32, device="cuda") for w, h in dims[1:]]
    pairs.append((d_in, d_base, lv))
    if k == 0:
        spacer = torch.empty(40 << 30, dtype=torch.uint8, device="cuda")
if not os.environ.get("PYR_PLACED"):
    del spacer
torch.cuda.empty_cache()
ALG = 7593190  # bytes per frame


def run(p):
    d_in, d_base, lv = p
    ctx.process_pyramid_batch(d_in.data_ptr(), d_base.data_ptr(), 4, [t.data_ptr() for t in lv], n, 15, s)


def timeit(p, reps=8):
    run(p)
    run(p)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(p)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


variants = [(c, t, f) for t in STREAMS for c in CHUNKS for f in FPBS]
res = {(v, k): [] for v in variants for k in range(len(pairs))}
for r in range(rounds):
    for v in variants:
        c, t, f = v
        ctx.set_option(capi.OPT_PREFETCH_STREAMS, t)
        ctx.set_option(capi.OPT_PREFETCH_CHUNK, c)
        ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, f)
        for k, p in enumerate(pairs):
            res[(v, k)].append(timeit(p))
print("config 5, %d frames per call, %d rounds; ms per call (median) and fraction of 8 TB/s of %d algorithmic bytes per frame" % (n, rounds, ALG))
for v in variants:
    line = "chunk %4s  streams %d  fpb %3s :" % (v[0] or "auto", v[1], v[2] or "auto")
    for k in range(len(pairs)):
        m = float(np.median(res[(v, k)]))
        line += "   pair %d  %.4f ms  %.4f" % (k, m, ALG * n / (m * 1e-3) / 8e12)
    print(line, flush=True)
