#!/usr/bin/env python3
"""The headline launch on buffers made by the product's allocator (mdc_alloc_placed_device), one strategy per process.

  python tools/placed_probe.py <strategy: first|malloc|vmm> [rounds] [frames]
      environment: MDC_PLACE_PIECE_MIB, MDC_PLACE_COMPOSE, MDC_PLACE_CANDIDATES, MDC_PLACE_SPREAD_MB

Per round: allocate, synthesise the frames, tune the plan (first round only), pre-roll 0.4 s, time 40 launches (HIP events on the launching
stream), spot-check two frames against the oracle, give the buffers back.  `rounds` > 1 is the allocator's soak: every round maps / unmaps
(vmm) or allocates / frees (malloc) everything again in one process.  Prints one PLACED line per round.
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from mono_dataset_code_amd import capi, synth  # noqa: E402
from oracle import loader  # noqa: E402

STRATEGY = sys.argv[1] if len(sys.argv) > 1 else "malloc"
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
FRAMES = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
NPI, NPO = 1280 * 1024, 640 * 480


def main():
    dev = torch.device("cuda", 0)
    d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_placed_"))
    with bench.quiet_stdout():
        fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
        photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
    ctx = capi.Context(0)
    ctx.bind(fov, photo)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    s = st.cuda_stream
    flags = capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED | capi.RECTIFY
    which = {"first": capi.PLACE_FIRST, "malloc": capi.PLACE_MALLOC, "vmm": capi.PLACE_VMM, "auto": capi.PLACE_AUTO}[STRATEGY]
    alg = (int(ctx.info().src_bbox_bytes) + NPO * 4) * FRAMES
    O = loader.Oracle()
    rx, ry = fov.remap()
    _, vinv = photo.vignette()
    ginv = photo.ginv()
    knobs = " ".join("%s=%s" % (k, os.environ[k]) for k in ("MDC_PLACE_PIECE_MIB", "MDC_PLACE_COMPOSE", "MDC_PLACE_CANDIDATES", "MDC_PLACE_SPREAD_MB") if k in os.environ)
    for rnd in range(ROUNDS):
        t0 = time.perf_counter()
        b = ctx.alloc_placed(FRAMES, flags, which, s)
        t_alloc = time.perf_counter() - t0
        ctx.synth_frames(b.d_in, 0, FRAMES, NPI, synth.SEED, s)
        if rnd == 0 and os.environ.get("PROBE_SHAPE"):  # e.g. 128x16:128 -- a given plan instead of the tuner's
            wh, fpb = os.environ["PROBE_SHAPE"].split(":")
            ctx.set_option(capi.OPT_TILE_COLS, int(wh.split("x")[0]))
            ctx.set_option(capi.OPT_TILE_ROWS, int(wh.split("x")[1]))
            ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, int(fpb))
            plan = "%s fpb %s (given)" % (wh, fpb)
        elif rnd == 0:
            t = ctx.tune(b.d_in, b.d_out, min(FRAMES, 4096), flags, s)
            plan = "%dx%d fpb %d" % (t.tile_w, t.tile_h, t.frames_per_block)
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < 0.4:
            for _ in range(20):
                ctx.process_batch(b.d_in, b.d_out, FRAMES, flags, s)
            torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
        for a_, b_ in ev:
            a_.record()
            ctx.process_batch(b.d_in, b.d_out, FRAMES, flags, s)
            b_.record()
        torch.cuda.synchronize()
        ms = np.array([a_.elapsed_time(b_) for a_, b_ in ev])
        bad = 0
        for f in sorted(set([0, FRAMES - 1] + list(range(rnd, FRAMES, 97)))):  # (every piece of both ranges is visited: a range lying on the wrong pages shows)
            raw = ctx.copy_to_host(b.d_in + f * NPI, NPI, np.uint8)
            want = O.get_image(raw, 1280, 1024, 640, 480, ginv, vinv, True, True, rx, ry, True, True, True, True)
            bad += bench.bits_differ(want, ctx.copy_to_host(b.d_out + f * NPO * 4, NPO, np.float32))
        desc = b.describe()
        t2 = time.perf_counter()
        ctx.free_placed(b)
        print("PLACED %-6s %s round %d: launch median %.4f ms min %.4f = frac %.4f | probe first %s chosen %s | alloc %.2f s free %.2f s | plan %s | mismatches %d | %s"
              % (STRATEGY, knobs, rnd, float(np.median(ms)), float(ms.min()), alg / (float(np.median(ms)) * 1e-3) / 8e12, desc["ms_on_first_allocations"],
                 desc["ms_on_chosen_pair"], t_alloc, time.perf_counter() - t2, plan, bad, desc["how"]), flush=True)
        if desc.get("ms_frames_i_results_j") and rnd == 0:
            for row in desc["ms_frames_i_results_j"]:
                print("   ", " ".join("%.4f" % x for x in row))


if __name__ == "__main__":
    main()
