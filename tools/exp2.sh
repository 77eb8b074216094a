#!/bin/bash
# GPU experiment batch 2: DVFS warm-up behaviour + PMC passes of the default fused kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp2; mkdir -p $O
python - > $O/warmup_curve.txt 2>&1 <<'PY'
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from mono_dataset_code_amd import capi, synth
d = synth.write_sequence_calibration(tempfile.mkdtemp())
so = os.dup(1); os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
import ctypes; ctypes.CDLL(None).fflush(None); os.dup2(so, 1)
ctx = capi.Context(0); ctx.bind(fov, photo)
B = 1024
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
d_in = torch.empty(B * 1280 * 1024, dtype=torch.uint8, device="cuda")
d_out = torch.empty(B * 640 * 480, dtype=torch.float32, device="cuda")
ctx.synth_frames(d_in.data_ptr(), 0, B, 1280 * 1024, synth.SEED, s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for rnd in range(60):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), B, 15, s)
    e1.record(); torch.cuda.synchronize()
    print("t=%.3fs round %d: %.4f ms/launch" % (time.perf_counter() - t0, rnd, e0.elapsed_time(e1) / 10), flush=True)
    if rnd == 30: time.sleep(2.0); print("-- slept 2 s")
PY
bash tools/pmc.sh exp2/pmc -- --frames 1024 > $O/pmc.txt 2>&1
cat $O/warmup_curve.txt $O/pmc.txt
