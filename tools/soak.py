#!/usr/bin/env python3
"""Soak test of the tiled kernel's hand-placed vmcnt barriers: many launches on fresh data, every
tile shape / window-buffer count, compared bit for bit with the gather kernel (which has no LDS
staging and no barrier).  Any race between the LDS-DMA and the tap reads shows up as a mismatch.
Every third iteration runs the FUSED PYRAMID on a 1280x1024 -> 1280x1024 remap and compares its levels with
the stand-alone level passes over the same base (the level-3 hand-over goes through LDS across the per-frame
barrier: a missing wait there shows up as a wrong level-3 pixel).
usage: python tools/soak.py [iterations]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mono_dataset_code_amd import capi, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
so = os.dup(1)
os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_soak_"))
fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
d2 = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_soak_pyr_"), synth.camera_lines(1280, 1024, 1280, 1024))
fov2 = capi.UndistorterFOV(os.path.join(d2, "camera.txt"))
import ctypes  # noqa: E402
ctypes.CDLL(None).fflush(None)
os.dup2(so, 1)
ctx = capi.Context(0)
ctx.bind(fov, photo)
pyr = capi.Context(0)
pyr.bind(fov2, photo)
B, npi, npo = 512, 1280 * 1024, 640 * 480
PB = 96  # frames of a pyramid launch (base 5.2 MB each)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
s = st.cuda_stream
d_in = torch.empty(B * npi, dtype=torch.uint8, device="cuda")
d_ref = torch.empty(B * npo, dtype=torch.float32, device="cuda")
d_out = torch.empty(B * npo, dtype=torch.float32, device="cuda")
p_base = torch.empty(PB * npi, dtype=torch.float32, device="cuda")
p_lv = [torch.empty(PB * (1280 >> l) * (1024 >> l), dtype=torch.float32, device="cuda") for l in (1, 2, 3)]
p_ref = [torch.empty_like(t) for t in p_lv]
configs = [(64, 32, 2), (64, 32, 3), (128, 16, 2), (64, 60, 2), (128, 32, 2), (64, 64, 2), (128, 16, 4), (64, 16, 2), (128, 32, 3), (64, 16, 4)]
# (two-stage selector, tile cols, rows, window buffers): 1 = wave-private strips (the automatic choice for this scale-1 remap),
# 2 = the direct kernel
pyr_configs = [(1, 0, 0, 2), (2, 128, 16, 0), (1, 0, 0, 1), (2, 64, 32, 2), (1, 0, 0, 3), (2, 128, 32, 2), (1, 0, 0, 2), (2, 64, 64, 3),
               (1, 0, 0, 4), (2, 64, 16, 4), (1, 0, 0, 1), (2, 128, 16, 2)]


def same(a, b):
    if torch.equal(a.view(torch.int32), b.view(torch.int32)):
        return True
    return torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))


bad = 0
for it in range(iters):
    ctx.synth_frames(d_in.data_ptr(), it * B, B, npi, synth.SEED + it, s)
    if it % 3 == 2:
        sel, cols, rows, nbuf = pyr_configs[(it // 3) % len(pyr_configs)]
        pyr.set_option(capi.OPT_TWO_STAGE, sel)
        pyr.set_option(capi.OPT_TILE_COLS, cols)
        pyr.set_option(capi.OPT_TILE_ROWS, rows)
        pyr.set_option(capi.OPT_WINDOW_BUFFERS, nbuf)
        pyr.set_option(capi.OPT_FRAMES_PER_BLOCK, (0, 5, 12)[it % 3])
        kname = pyr.describe_launch(15, 4)
        # (the 1024-thread tiles have no fused-pyramid instantiation: their levels come from the per-level passes)
        assert kname.startswith({1: "remap_strip_kernel<true, true", 2: "remap_tiled_kernel<true, false, %s" % ("true" if cols * rows <= 2048 else "false")}[sel]), kname
        for rep in range(3):
            for t in p_lv:
                t.fill_(-7.0)
            pyr.process_pyramid_batch(d_in.data_ptr(), p_base.data_ptr(), 4, [t.data_ptr() for t in p_lv], PB, 15, s)
            pyr.pyramid_batch(p_base.data_ptr(), 1280, 1024, 4, [t.data_ptr() for t in p_ref], PB, s)
            torch.cuda.synchronize()
            for l, (a, b) in enumerate(zip(p_lv, p_ref)):
                if not same(a, b):
                    bad += 1
                    print("PYRAMID MISMATCH iteration", it, "kernel", kname, "tile", cols, rows, "nbuf", nbuf, "level", l + 1, "rep", rep, flush=True)
        continue
    ctx.set_option(capi.OPT_KERNEL, capi.KERNEL_GATHER)
    ctx.process_batch(d_in.data_ptr(), d_ref.data_ptr(), B, 15, s)
    ctx.set_option(capi.OPT_KERNEL, capi.KERNEL_TILED)
    cols, rows, nbuf = configs[it % len(configs)]
    ctx.set_option(capi.OPT_TILE_COLS, cols)
    ctx.set_option(capi.OPT_TILE_ROWS, rows)
    ctx.set_option(capi.OPT_WINDOW_BUFFERS, nbuf)
    ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, (0, 7, 33)[it % 3])
    for rep in range(4):
        d_out.fill_(-7.0)
        ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), B, 15, s)
        torch.cuda.synchronize()
        if not same(d_out, d_ref):
            bad += 1
            print("MISMATCH iteration", it, "tile", cols, rows, "nbuf", nbuf, "rep", rep, flush=True)
print("soak: %d iterations (2/3 fused x 4 launches x %d frames, 1/3 fused pyramid x 3 launches x %d frames), %d mismatching launches"
      % (iters, B, PB, bad))
sys.exit(1 if bad else 0)
