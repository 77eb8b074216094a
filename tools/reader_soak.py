#!/usr/bin/env python3
"""Randomised soak of the reader's JPEG path (stage 2: Huffman decoding on the device, three-stream pipeline, slab pool, results
made ahead): for SECONDS seconds random getImages ranges (1..700 frames, all four switch combinations that rectify or not) and
random getImage walks (runs in order, jumps, switch changes) on a zipped sequence of 700 small JPEGs, every result compared bit
for bit with the host-decoded path (stage 0, lookahead off) computed once.  usage: python tools/reader_soak.py [seconds] [seed]"""
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


r = capi.DatasetReader(d)
calls = frames_done = singles = 0
t0 = time.time()
while time.time() - t0 < SECONDS:
    sw = SWITCHES[rng.integers(len(SWITCHES))]
    if rng.random() < 0.5:
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
print("READER_SOAK ok: %.0f s, %d getImages calls (%d frames), %d getImage calls, every result bit-identical to the host-decoded path"
      % (time.time() - t0, calls, frames_done, singles))
