#!/usr/bin/env python3
"""Why does the same launch run 1.61 ms now and 1.49 ms fifteen seconds later in one process (BENCH_r04)?

The headline launch (4096 frames, tuned plan) is repeated back to back while the GPU's clocks, power and temperature are
read from sysfs every 0.25 s:
  phase A   from a cold start: SECONDS_A of launches
  phase B   IDLE_S idle, then SECONDS_B of launches                    (does an idle gap undo the ramp?)
  phase C   fresh input / output allocations, SECONDS_B of launches    (physical placement)
  phase D   the linear-mix ceiling kernel alone, SECONDS_B             (does the memory system itself drift?)
One line per 0.25-s bin: mean / min launch time, sclk / mclk / fclk / socclk (the starred pp_dpm level), power, temperatures.
  python tools/clock_probe.py [SECONDS_A [SECONDS_B [IDLE_S]]]
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402  (gpu_clock_snapshot, quiet_stdout)
from mono_dataset_code_amd import capi, synth  # noqa: E402

A_S = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
B_S = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
IDLE_S = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
FRAMES = int(os.environ.get("PROBE_FRAMES", "4096"))
NPIX_IN, NPIX_OUT = 1280 * 1024, 640 * 480


def run_phase(name, launch, seconds, t_origin):
    """Launches back to back for `seconds`; a sysfs snapshot per 0.25 s; -> rows (t, mean ms, min ms, n, snapshot)."""
    rows = []
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        evs = []
        t_bin = time.perf_counter()
        while time.perf_counter() - t_bin < 0.25:
            for _ in range(8):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                launch()
                b.record()
                evs.append((a, b))
            snap = bench.gpu_clock_snapshot(0)  # while the 8 launches are in flight: the clocks under load, not after the queue drained
            torch.cuda.synchronize()
        ms = np.array([a.elapsed_time(b) for a, b in evs])
        rows.append((time.perf_counter() - t_origin, float(ms.mean()), float(ms.min()), len(ms), snap))
    print("--- phase %s" % name)
    for t, mean, mn, n, snap in rows:
        print("t %6.2f s  launch mean %.4f min %.4f ms (%3d)  %s" % (t, mean, mn, n, bench.format_clock_snapshot(snap)))
    sys.stdout.flush()
    return rows


def main():
    dev = torch.device("cuda", 0)
    d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_probe_"))
    with bench.quiet_stdout():
        fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
        photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
    ctx = capi.Context(0)
    ctx.bind(fov, photo)
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    s = tstream.cuda_stream
    flags = capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED | capi.RECTIFY

    def buffers():
        d_in = torch.empty(FRAMES * NPIX_IN, dtype=torch.uint8, device=dev)
        d_out = torch.empty(FRAMES * NPIX_OUT, dtype=torch.float32, device=dev)
        ctx.synth_frames(d_in.data_ptr(), 0, FRAMES, NPIX_IN, synth.SEED, s)
        torch.cuda.synchronize()
        return d_in, d_out

    print("idle snapshot before anything ran: %s" % bench.format_clock_snapshot(bench.gpu_clock_snapshot(0)))
    d_in, d_out = buffers()
    t = ctx.tune(d_in.data_ptr(), d_out.data_ptr(), FRAMES, flags, s)
    print("plan: tile %dx%d, %d frames per workgroup, tuner %.4f ms; kernel %s" % (t.tile_w, t.tile_h, t.frames_per_block, t.ms,
                                                                                     ctx.describe_launch(flags, 0)))
    t0 = time.perf_counter()
    rows = run_phase("A: cold start, %g s of launches" % A_S, lambda: ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), FRAMES, flags, s), A_S, t0)
    time.sleep(IDLE_S)
    rows_b = run_phase("B: after %g s idle" % IDLE_S, lambda: ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), FRAMES, flags, s), B_S, t0)
    keep = (d_in, d_out)  # the old allocations stay alive: the new ones cannot land on the same physical pages
    d_in2, d_out2 = buffers()
    rows_c = run_phase("C: fresh allocations", lambda: ctx.process_batch(d_in2.data_ptr(), d_out2.data_ptr(), FRAMES, flags, s), B_S, t0)
    info = ctx.info()
    rb, wb = int(info.src_bbox_bytes) * FRAMES // 16 * 16, NPIX_OUT * 4 * FRAMES
    run_phase("D: linear-mix ceiling kernel alone", lambda: ctx.ceiling_mix(d_in.data_ptr(), rb, d_out.data_ptr(), wb, 65536, 0, s), B_S, t0)
    del keep

    def med(rs, lo, hi):
        v = [r[1] for r in rs if lo <= r[0] - rs[0][0] < hi]
        return float(np.median(v)) if v else float("nan")

    print("=== summary (median of the bins' mean launch time, ms)")
    print("A first 0.5 s %.4f | 0.5-1 s %.4f | 1-2 s %.4f | 2-4 s %.4f | 4-8 s %.4f | 8+ s %.4f" %
          (med(rows, 0, 0.5), med(rows, 0.5, 1), med(rows, 1, 2), med(rows, 2, 4), med(rows, 4, 8), med(rows, 8, 1e9)))
    print("B first 0.5 s %.4f | rest %.4f      C first 0.5 s %.4f | rest %.4f" %
          (med(rows_b, 0, 0.5), med(rows_b, 0.5, 1e9), med(rows_c, 0, 0.5), med(rows_c, 0.5, 1e9)))


if __name__ == "__main__":
    main()
