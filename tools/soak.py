#!/usr/bin/env python3
"""Soak test of the tiled kernel's hand-placed vmcnt barriers: many launches on fresh data, every
tile shape / window-buffer count, compared bit for bit with the gather kernel (which has no LDS
staging and no barrier).  Any race between the LDS-DMA and the tap reads shows up as a mismatch.
usage: python tools/soak.py [iterations]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mono_dataset_code_amd import capi, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_soak_"))
so = os.dup(1)
os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
import ctypes  # noqa: E402
ctypes.CDLL(None).fflush(None)
os.dup2(so, 1)
ctx = capi.Context(0)
ctx.bind(fov, photo)
B, npi, npo = 512, 1280 * 1024, 640 * 480
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
s = st.cuda_stream
d_in = torch.empty(B * npi, dtype=torch.uint8, device="cuda")
d_ref = torch.empty(B * npo, dtype=torch.float32, device="cuda")
d_out = torch.empty(B * npo, dtype=torch.float32, device="cuda")
configs = [(32, 2), (32, 3), (32, 4), (60, 2), (60, 3), (64, 2), (64, 3), (16, 2), (16, 4)]
bad = 0
for it in range(iters):
    ctx.synth_frames(d_in.data_ptr(), it * B, B, npi, synth.SEED + it, s)
    ctx.set_option(capi.OPT_KERNEL, capi.KERNEL_GATHER)
    ctx.process_batch(d_in.data_ptr(), d_ref.data_ptr(), B, 15, s)
    ctx.set_option(capi.OPT_KERNEL, capi.KERNEL_TILED)
    rows, nbuf = configs[it % len(configs)]
    ctx.set_option(capi.OPT_TILE_ROWS, rows)
    ctx.set_option(capi.OPT_WINDOW_BUFFERS, nbuf)
    ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, (0, 7, 33)[it % 3])
    for rep in range(4):
        d_out.fill_(-7.0)
        ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), B, 15, s)
        torch.cuda.synchronize()
        if not torch.equal(d_out.view(torch.int32), d_ref.view(torch.int32)):
            nan_ok = torch.equal(torch.isnan(d_out), torch.isnan(d_ref)) and torch.equal(torch.nan_to_num(d_out), torch.nan_to_num(d_ref))
            if not nan_ok:
                bad += 1
                print("MISMATCH iteration", it, "rows", rows, "nbuf", nbuf, "rep", rep, flush=True)
print("soak: %d iterations x 4 launches x %d frames, %d mismatching launches" % (iters, B, bad))
sys.exit(1 if bad else 0)
