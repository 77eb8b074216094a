#!/usr/bin/env python3
"""vignetteCalib solver half-iterations (reference src/main_vignetteCalib.cpp:395-527) at the tool's real size --
gw x gh = 1000 x 1000 plane points (:122-123), 1280 x 1024 images -- on the GPU (all images resident in HBM) and the
reference's own loops on one host core (as shipped; measured on fewer images, it is linear in n).
usage: python tools/vcal_rate.py [n_images_gpu] [n_images_cpu]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mono_dataset_code_amd import capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 6
wI, hI, gw, gh = 1280, 1024, 1000, 1000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
yy, xx = torch.meshgrid(torch.arange(gh, device=dev, dtype=torch.float32), torch.arange(gw, device=dev, dtype=torch.float32), indexing="ij")
images = torch.empty((N, hI, wI), dtype=torch.float32, device=dev)
p2x = torch.empty((N, gw * gh), dtype=torch.float32, device=dev)
p2y = torch.empty((N, gw * gh), dtype=torch.float32, device=dev)
for i in range(N):
    sx, sy = 0.9 + 0.3 * torch.rand(2, generator=g, device=dev)
    ox, oy = 2 + 60 * torch.rand(2, generator=g, device=dev)
    px, py = ox + sx * xx + 0.03 * yy, oy + 0.85 * sy * yy - 0.02 * xx
    bad = (px < 1) | (py < 1) | (px > wI - 2.5) | (py > hI - 2.5)
    px[bad] = float("nan")
    py[bad] = float("nan")
    p2x[i], p2y[i] = px.reshape(-1), py.reshape(-1)
    images[i] = 80 + 40 * torch.rand((hI, wI), generator=g, device=dev)
ctx = capi.Context(0)
st = torch.cuda.current_stream().cuda_stream
pc = torch.zeros(gw * gh, dtype=torch.float32, device=dev)
vf = torch.ones(hI * wI, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
index = ctx.vcal_index(images, p2x, p2y, st)
torch.cuda.synchronize()
print("GPU contribution index: built in %.1f ms, %.2f GB for %.1f M list entries (%.1f %% padding)"
      % ((time.perf_counter() - t0) * 1e3, index.bytes / 1e9, index.entries / 1e6, 100.0 * (index.bytes / 16.0 / max(index.entries, 1) - 1)),
      flush=True)
for name, fn in (("plane step", lambda: ctx.vcal_plane_step(images, p2x, p2y, pc, vf, 10 ** 8, st)),
                 ("vignette step, atomics", lambda: ctx.vcal_vignette_step(images, p2x, p2y, pc, vf, 10 ** 8, st)),
                 ("vignette step, indexed", lambda: ctx.vcal_vignette_step_indexed(index, pc, vf, 10 ** 8, st))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    samples = N * gw * gh
    print("GPU %-22s: %d images, %.1f ms per half-iteration = %.2f G samples/s" % (name, N, dt * 1e3, samples / dt / 1e9), flush=True)

# the per-image preparation that has a device form: gradient mask (:293-301, wavefront replay of the sequential loop),
# coordinate mask (:345-357), output smoothing (:541-566)
work = images.clone()
torch.cuda.synchronize()
t0 = time.perf_counter()
ctx.vcal_gradient_mask(work, 255, st)
torch.cuda.synchronize()
print("GPU gradient mask         : %d images %dx%d in %.1f ms (one workgroup per image, %d barrier steps)"
      % (N, wI, hI, (time.perf_counter() - t0) * 1e3, (wI - 3) + 5 * (hI - 3) - 11), flush=True)
cx, cy = p2x.clone(), p2y.clone()
torch.cuda.synchronize()
t0 = time.perf_counter()
ctx.vcal_mask_coords(cx, cy, wI, hI, st)
torch.cuda.synchronize()
print("GPU coordinate mask       : %d x %d points in %.2f ms" % (N, gw * gh, (time.perf_counter() - t0) * 1e3), flush=True)
t0 = time.perf_counter()
ctx.vcal_smooth(vf, wI, hI, st)
torch.cuda.synchronize()
print("GPU output smoothing      : %dx%d in %.2f ms" % (wI, hI, (time.perf_counter() - t0) * 1e3), flush=True)
del work, cx, cy

try:
    from oracle import loader

    if NC <= 0:
        raise OSError("skipped (0 images asked for)")
    ref = loader.VcalRef()
    im = images[:NC].cpu().numpy()
    x, y = p2x[:NC].cpu().numpy(), p2y[:NC].cpu().numpy()
    pcn, vfn = np.zeros(gw * gh, np.float32), np.ones(hI * wI, np.float32)
    for name, fn in (("plane step", lambda: ref.plane_step(im, x, y, gw, gh, pcn, vfn, 10 ** 8)),
                     ("vignette step", lambda: ref.vignette_step(im, x, y, gw, gh, pcn, vfn, 10 ** 8))):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        print("CPU reference %-14s: %d images, %.1f ms = %.3f G samples/s (one thread, as shipped)" % (name, NC, dt * 1e3, NC * gw * gh / dt / 1e9), flush=True)
except OSError as e:
    print("reference loops not available:", e)
