#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_47; mkdir -p $O
timeout 600 python tools/exp.py --out 640x480 --frames 1 --pyramid 0 --cols 0,64,128 --rows 0,8,16,32 --rounds 3 --iters 50 2>&1 | grep -v amdgpu.ids > $O/one_frame_shapes.txt; cat $O/one_frame_shapes.txt
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
pin_in = capi.PinnedArray((W * H,), np.uint8); pin_out = capi.PinnedArray((w * h,), np.float32)
pin_in.array[:] = synth.noise_frames(0, 1, W * H)[0]
d_in = torch.from_numpy(pin_in.array.copy()).cuda()
st = torch.cuda.Stream(); s = st.cuda_stream
for cols, rows in ((0, 0), (128, 32), (128, 16), (64, 32), (64, 16), (128, 8), (64, 8)):
    ctx.set_option(capi.OPT_TILE_COLS, cols); ctx.set_option(capi.OPT_TILE_ROWS, rows)
    def run_host():
        ctx.process_host(pin_in.array, pin_out.array, 15)
    def run_dev_to_host():
        ctx.process_batch(d_in.data_ptr(), pin_out.array.ctypes.data, 1, 15, s); st.synchronize()
    for fn, name in ((run_host, "mdc_process_host (zero copy both ways)"), (run_dev_to_host, "device frame -> host image")):
        fn(); t0 = time.perf_counter()
        for _ in range(300): fn()
        t = (time.perf_counter() - t0) / 300
        print("tile %3d x %2d  %-40s %6.1f us" % (cols, rows, name, t * 1e6), ctx.describe_launch(15, 0))
P
