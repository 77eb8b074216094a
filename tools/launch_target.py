#!/usr/bin/env python3
"""N launches of the fused kernel (or the fused-pyramid one) on 1024 frames: the process diagnosis builds and profilers are
pointed at (tools/phase_timing.sh; PC sampling and thread trace are not available on the pool's boxes: rocprofv3 reports no
agent that supports PC sampling, and the ATT decoder library is not installed).
usage: python tools/launch_target.py [fused|pyramid|base1280] [launches] [two-stage selector 0/1/2]
(base1280 = the scale-1 rectification of the pyramid workload without the levels)"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mono_dataset_code_amd import capi, synth  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "fused"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cam = None if what == "fused" else synth.camera_lines(1280, 1024, 1280, 1024)
d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_pcs_"), cam) if cam else synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_pcs_"))
fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
ctx = capi.Context(0)
ctx.bind(fov, photo)
if len(sys.argv) > 3:
    ctx.set_option(capi.OPT_TWO_STAGE, int(sys.argv[3]))
n = 1024
npi = 1280 * 1024
info = ctx.info()
npo = info.out_w * info.out_h
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
s = st.cuda_stream
d_in = torch.empty(n * npi, dtype=torch.uint8, device="cuda")
ctx.synth_frames(d_in.data_ptr(), 0, n, npi, synth.SEED, s)
d_out = torch.empty(n * npo, dtype=torch.float32, device="cuda")
lv = [torch.empty(n * (npo >> (2 * l)), dtype=torch.float32, device="cuda") for l in (1, 2, 3)] if what == "pyramid" else []
for _ in range(launches):
    if what == "pyramid":
        ctx.process_pyramid_batch(d_in.data_ptr(), d_out.data_ptr(), 4, [t.data_ptr() for t in lv], n, 15, s)
    else:
        ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), n, 15, s)
torch.cuda.synchronize()
print("done", what, launches)
