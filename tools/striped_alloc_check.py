#!/usr/bin/env python3
"""mdc_device_alloc of a GiB or more and mdc_alloc_striped_set_device hand out ranges assembled from physical pieces: frames written by a kernel through such a
range must come back through hipMemcpy, in every piece, after earlier ranges of the process were freed (the address-range hazard of
tools/vmm_offset_check.hip).   python tools/striped_alloc_check.py"""
import os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mono_dataset_code_amd import capi, synth
d = synth.write_sequence_calibration(tempfile.mkdtemp())
fov = capi.UndistorterFOV(os.path.join(d, "camera.txt")); photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
ctx = capi.Context(0); ctx.bind(fov, photo)
npix = 1280*1024
for n in (900, 1700):
    p = ctx.device_alloc(n*npix)
    print("ptr %x bytes %d" % (p, n*npix), flush=True)
    ctx.synth_frames(p, 5, n, npix, synth.SEED, 0)
    torch.cuda.synchronize()
    for f in (0, 1, n//3, n//2, n-1):
        raw = ctx.copy_to_host(p + f*npix, npix, np.uint8)
        ok = np.array_equal(raw, synth.noise_frames(5+f, 1, npix)[0])
        print(" frame", f, "ok" if ok else "BAD", raw[:4], raw[-4:], "offset MiB", f*npix/2**20, flush=True)
    ctx.device_free(p)
b = ctx.alloc_striped_set([900*npix], 0)
print("set", b.note.decode())
ctx.synth_frames(b.d_ptr[0], 5, 900, npix, synth.SEED, 0); torch.cuda.synchronize()
for f in (0, 450, 899):
    raw = ctx.copy_to_host(b.d_ptr[0] + f*npix, npix, np.uint8)
    print(" set frame", f, np.array_equal(raw, synth.noise_frames(5+f, 1, npix)[0]))
