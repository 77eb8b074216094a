#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python - <<'P' 2>&1 | grep -v "amdgpu.ids\|^Input\|^Out\|resolution\|Reading\|Success"
import os, sys, tempfile, time
sys.path.insert(0, ".")
import numpy as np, torch
from mono_dataset_code_amd import capi, synth
d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_zc_"))
fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
ctx = capi.Context(0); ctx.bind(fov, photo)
W, H, w, h = fov.dims()
N = 64
d_in = torch.from_numpy(np.stack(synth.noise_frames(0, N, W * H))).cuda()
one = capi.PinnedArray((N, w * h), np.float32)
many = [capi.PinnedArray((w * h,), np.float32) for _ in range(N)]
big2 = capi.PinnedArray((N, w * h + 4096), np.float32)  # one block, rows not back to back
st = torch.cuda.Stream(); s = st.cuda_stream
st2 = torch.cuda.Stream(); s2 = st2.cuda_stream
def per_frame(ptrs, two=False):
    for i in range(N):
        ctx.process_batch(d_in[i].data_ptr(), ptrs[i], 1, 15, s2 if (two and i & 1) else s)
    st.synchronize(); st2.synchronize()
def timeit(fn):
    fn(); fn(); t0 = time.perf_counter()
    for _ in range(10): fn()
    return (time.perf_counter() - t0) / 10 / N * 1e6
p_one = [one.array[i].ctypes.data for i in range(N)]
p_many = [m.array.ctypes.data for m in many]
p_big2 = [big2.array[i].ctypes.data for i in range(N)]
print("one launch of 64 frames into one block           : %.1f us per frame" % timeit(lambda: (ctx.process_batch(d_in.data_ptr(), p_one[0], N, 15, s), st.synchronize())))
print("64 single-frame launches, rows of one block      : %.1f us per frame" % timeit(lambda: per_frame(p_one)))
print("64 single-frame launches, rows of one block, gaps: %.1f us per frame" % timeit(lambda: per_frame(p_big2)))
print("64 single-frame launches, 64 separate allocations: %.1f us per frame" % timeit(lambda: per_frame(p_many)))
print("... alternating over two streams, one block      : %.1f us per frame" % timeit(lambda: per_frame(p_one, True)))
print("... alternating over two streams, 64 allocations : %.1f us per frame" % timeit(lambda: per_frame(p_many, True)))
P
