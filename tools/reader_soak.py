#!/usr/bin/env python3
"""Randomised soak of the reader's JPEG path (stage 2: Huffman decoding on the device, three-stream pipeline, slab pool, results
made ahead): for SECONDS seconds random getImages ranges (1..700 frames, all four switch combinations that rectify or not) and
random getImagesDevice ranges (results, box levels and gradient images left in device arrays; two lanes on the device) and random getImage walks
(runs in order, jumps, switch changes) on a zipped sequence of 700 small JPEGs, every result compared bit
for bit with the host-decoded path (stage 0, lookahead off) computed once.  usage: python tools/reader_soak.py [seconds] [seed]
SOAK_DEVICE_SHARE=0.8: share of getImagesDevice calls in the mix (default 0.25).  SOAK_NO_SYNC=1 leaves out the caller's own
synchronisation before a getImagesDevice call -- the fill of the arrays then races the results (profiles/r05_soaks.txt)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from mono_dataset_code_amd import capi  # noqa: E402
from test_reader_cpu import make_sequence, textured  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
H, W, N = 96, 160, 700
d = tempfile.mkdtemp(prefix="mdc_reader_soak_")
base = [textured(H, W, s) for s in range(16)]
frames = [np.roll(base[i % 16], i, 1) for i in range(N)]
frames[5][:] = 255
make_sequence(d, frames, True, "jpg")
SWITCHES = [(1, 1, 1, 1), (0, 1, 1, 0), (1, 0, 0, 1), (0, 0, 0, 0)]
ref = capi.DatasetReader(d)
ref.set_gpu_jpeg(0)
ref.set_lookahead(0)
want = {}
for sw in SWITCHES:
    imgs, ok, n = ref.get_images(0, N, *sw)
    assert n == N and ok.all()
    want[sw] = imgs.copy()
ref.close()


def same(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


import torch  # noqa: E402

r = capi.DatasetReader(d)
OW, OH = r.out_w, r.out_h
dev_calls = 0
d_arrays = {}  # rectified? -> device arrays for the whole sequence


def device_arrays(rect):
    if rect not in d_arrays:
        w0, h0 = (OW, OH) if rect else (W, H)
        dims = [(w0 >> l, h0 >> l) for l in range(3)]
        d_arrays[rect] = (torch.zeros((N, w0 * h0), dtype=torch.float32, device="cuda"),
                          [torch.zeros((N, a * b), dtype=torch.float32, device="cuda") for a, b in dims[1:]],
                          [torch.zeros((N, a * b * 3), dtype=torch.float32, device="cuda") for a, b in dims],
                          [torch.zeros((N, a * b), dtype=torch.float32, device="cuda") for a, b in dims], dims)
    return d_arrays[rect]


def explain(got, base, first, count, sw, with_levels):
    """a getImagesDevice mismatch: which positions, holding what (the fill value, another frame, a result that arrives late)"""
    bad = [i for i in range(count) if not same(got[i], want[sw][first + i])]
    print("MISMATCH getImagesDevice(first %d, count %d, switches %s, levels %s) after %d device calls: %d positions differ: %s" %
          (first, count, sw, with_levels, dev_calls, len(bad), bad[:20]), flush=True)
    time.sleep(0.05)
    torch.cuda.synchronize()
    late = base[:count].cpu().numpy()
    for i in bad[:8]:
        g = got[i]
        nfill = int((g == -1.0).sum())
        ndiff = int((g.view(np.uint32) != want[sw][first + i].view(np.uint32)).sum())
        where = [(s2, j) for s2 in SWITCHES for j in range(N) if same(g, want[s2][j])][:3]
        idx = np.flatnonzero(g.view(np.uint32) != want[sw][first + i].view(np.uint32))
        print("  position %d (frame %d): %d of %d values differ (first at %d, last at %d), %d hold the fill value; equal to (switches, frame) %s; 50 ms later %s" %
              (i, first + i, ndiff, g.size, idx[0], idx[-1], nfill, where, "equal" if same(late[i], want[sw][first + i]) else "still different"), flush=True)


DEVICE_SHARE = float(os.environ.get("SOAK_DEVICE_SHARE", "0.25"))
calls = frames_done = singles = 0
t0 = time.time()
while time.time() - t0 < SECONDS:
    sw = SWITCHES[rng.integers(len(SWITCHES))]
    u0 = rng.random()
    if u0 < DEVICE_SHARE:  # getImagesDevice: positions 0..count-1 of the arrays hold frames first..first+count-1
        count = int(rng.choice([1, 63, 64, 65, 129, 256, 257, 511, 700, int(rng.integers(1, 701))]))
        first = int(rng.integers(0, N - count + 1))
        base, lv, dI, ab, dims = device_arrays(sw[0])
        base.fill_(-1.0)
        if not os.environ.get("SOAK_NO_SYNC"):
            torch.cuda.synchronize()  # the call runs on the reader's own streams: the caller's pending work on the arrays must be done (include/mdc_hip.h)
        with_levels = rng.random() < 0.5
        outs = capi.DeviceOutputs.make(base.data_ptr(), 3, [t.data_ptr() for t in lv], [t.data_ptr() for t in dI], [t.data_ptr() for t in ab]) if with_levels \
            else capi.DeviceOutputs.make(base.data_ptr())
        valid, n = r.get_images_device(first, count, *sw, outs)
        assert n == count and valid.all(), (first, count, sw, n, r.last_error())
        got = base[:count].cpu().numpy()
        if not same(got, want[sw][first:first + count]):
            explain(got, base, first, count, sw, with_levels)
            raise AssertionError(("getImagesDevice", first, count, sw))
        if with_levels:  # level 1 of a few frames: the box filter of the base, bit for bit (0.25f * (((a + b) + c) + d))
            for f in (0, count - 1):
                b0 = want[sw][first + f].reshape(dims[0][1], dims[0][0])
                l1 = np.float32(0.25) * (((b0[0::2, 0::2][:dims[1][1], :dims[1][0]] + b0[0::2, 1::2][:dims[1][1], :dims[1][0]]) + b0[1::2, 0::2][:dims[1][1], :dims[1][0]]) + b0[1::2, 1::2][:dims[1][1], :dims[1][0]])
                g = lv[0][f].cpu().numpy().reshape(dims[1][1], dims[1][0])
                nan = np.isnan(l1)
                assert np.array_equal(nan, np.isnan(g)) and np.array_equal(l1[~nan].view(np.uint32), g[~nan].view(np.uint32)), ("level 1", first, f)
        dev_calls += 1
        frames_done += count
    elif u0 < max(0.6, DEVICE_SHARE + 0.05):
        count = int(rng.choice([1, 2, 63, 64, 65, 128, 255, 256, 257, 300, 512, 513, 700, int(rng.integers(1, 701))]))
        first = int(rng.integers(0, N - count + 1))
        imgs, ok, n = r.get_images(first, count, *sw)
        assert n == count and ok.all(), (first, count, sw, n)
        assert same(imgs, want[sw][first:first + count]), ("getImages", first, count, sw)
        calls += 1
        frames_done += count
    else:
        i = int(rng.integers(0, N))
        for _ in range(int(rng.integers(1, 400))):
            got = r.get_image(i, *sw)
            assert got is not None and got[3] == i, (i, sw)
            assert same(got[0].reshape(-1), want[sw][i]), ("getImage", i, sw)
            singles += 1
            u = rng.random()
            if u < 0.9:
                i = (i + 1) % N
            elif u < 0.95:
                i = int(rng.integers(0, N))
            else:
                sw = SWITCHES[rng.integers(len(SWITCHES))]
r.close()
print("READER_SOAK ok: %.0f s, %d getImages + %d getImagesDevice calls (%d frames), %d getImage calls, every result bit-identical to the host-decoded path"
      % (time.time() - t0, calls, dev_calls, frames_done, singles))
