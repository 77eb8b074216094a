#!/usr/bin/env python3
"""The headline launch against TWO linear streams on the same pair of buffers: one of its ALGORITHMIC bytes (bounding box + results: what
`roofline.frac_of_same_box_mix_ceiling` uses) and one of the bytes it really moves over the fabric (profiles/hbm_traffic.json: 0.99 MB read
+ 1.23 MB written per frame).  Is the launch at the memory system's rate for its traffic, i.e. is the read over-fetch what separates it
from the algorithmic stream?   python tools/traffic_ceiling.py [pairs]
"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from mono_dataset_code_amd import capi, synth  # noqa: E402

PAIRS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
FRAMES, NPI, NPO = 4096, 1280 * 1024, 640 * 480


def timeit(fn, reps=10):
    fn()
    fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main():
    d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_tc_"))
    with bench.quiet_stdout():
        fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
        photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
    ctx = capi.Context(0)
    ctx.bind(fov, photo)
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    s = st.cuda_stream
    flags = capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED | capi.RECTIFY
    info = ctx.info()
    name = ctx.describe_launch(flags, 0)
    tr = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get("fused:" + name, {})
    rd_meas = float(tr.get("read_bytes_per_frame", 991786.0))
    alg_r, alg_w = int(info.src_bbox_bytes), NPO * 4
    print("%s; per frame: algorithmic %d B read + %d B written; measured over the fabric %.0f B read (%s)" % (name, alg_r, alg_w, rd_meas, tr.get("source", "default")))
    pairs = []
    for k in range(PAIRS):
        a = torch.empty(FRAMES * NPI, dtype=torch.uint8, device="cuda")
        ctx.synth_frames(a.data_ptr(), 0, FRAMES, NPI, synth.SEED, s)
        b = torch.empty(FRAMES * NPO, dtype=torch.float32, device="cuda")
        pairs.append((a, b))
        if k + 1 < PAIRS:
            spacer = torch.empty(30 << 30, dtype=torch.uint8, device="cuda")
    del spacer
    torch.cuda.empty_cache()
    t = ctx.tune(pairs[0][0].data_ptr(), pairs[0][1].data_ptr(), FRAMES, flags, s)
    print("plan: tile %d x %d, %d frames per workgroup" % (t.tile_w, t.tile_h, t.frames_per_block))
    for k, (a, b) in enumerate(pairs):
        kern = timeit(lambda: ctx.process_batch(a.data_ptr(), b.data_ptr(), FRAMES, flags, s))
        line = "pair %d: launch %.4f ms" % (k, kern)
        for tag, rb in (("algorithmic", alg_r * FRAMES // 16 * 16), ("measured traffic", int(rd_meas * FRAMES) // 16 * 16)):
            best = None
            for blocks in (4096, 16384, 65536):
                for span in (0, 1):
                    c = timeit(lambda: ctx.ceiling_mix(a.data_ptr(), rb, b.data_ptr(), alg_w * FRAMES, blocks, span, s), reps=6)
                    best = c if best is None else min(best, c)
            line += ";  linear stream of the %s bytes %.4f ms (launch at %.3f of it)" % (tag, best, best / kern)
        print(line, flush=True)


if __name__ == "__main__":
    main()
