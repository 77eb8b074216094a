#!/usr/bin/env python3
"""UndistorterFOV::distortCoordinates as vignetteCalib calls it (reference src/main_vignetteCalib.cpp:284: 10^6 plane points per
image, one call per image): the unchanged class method, which sends bulk calls (n >= 65536) to the device, against the same
method kept on the host (MDC_DISTORT_GPU_MIN=0), same process image, bytes compared.
usage: python tools/distort_rate.py [points per call = 1000000] [calls = 50]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(n, calls, out):
    import numpy as np

    from mono_dataset_code_amd import capi, synth

    d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_dist_"))
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    rng = np.random.RandomState(3)
    x0 = (rng.rand(n).astype(np.float32) * 1.4 - 0.2) * np.float32(640)
    y0 = (rng.rand(n).astype(np.float32) * 1.4 - 0.2) * np.float32(480)
    x, y = x0.copy(), y0.copy()
    fov.distort_coordinates(x, y)  # first call: context, staging buffers
    t = []
    for _ in range(calls):
        x[:], y[:] = x0, y0
        t0 = time.perf_counter()
        fov.distort_coordinates(x, y)
        t.append(time.perf_counter() - t0)
    np.save(out, np.stack([x, y]))
    t.sort()
    sys.stderr.write("RATE median %.3f ms per call of %d points (%.1f M points/s), min %.3f ms\n" % (t[len(t) // 2] * 1e3, n, n / t[len(t) // 2] / 1e6, t[0] * 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    import numpy as np

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    tmp = tempfile.mkdtemp(prefix="mdc_dist_out_")
    res = {}
    for name, env in (("host loop (MDC_DISTORT_GPU_MIN=0)", {"MDC_DISTORT_GPU_MIN": "0"}), ("class method as shipped (device for n >= 65536)", {})):
        out = os.path.join(tmp, "%d.npy" % len(res))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), str(calls), out], env=dict(os.environ, **env),
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        line = [l for l in r.stderr.splitlines() if l.startswith("RATE")]
        print("%-50s %s" % (name, line[-1][5:] if line else "FAILED rc=%d %s" % (r.returncode, r.stderr[-300:])), flush=True)
        res[name] = np.load(out) if os.path.exists(out) else None
    a, b = list(res.values())
    if a is not None and b is not None:
        print("results bitwise equal:", bool(np.array_equal(a.view(np.uint32), b.view(np.uint32))))
