#!/usr/bin/env python3
"""Which pieces of a device's memory are slow WITH which?

profiles/r05_experiments/10_*: a (frames, results) pair of buffers is slow when both are of one "class" of device memory.  This probe allocates
N pieces of PIECE_GB (hipMalloc, in a row: neighbouring pieces are mostly neighbours in memory) and times a linear stream that reads one
piece and writes another (libmdc_bench: mdcb_ceiling_mix_device, equal byte counts) for a fixed reference piece against every piece, in
both roles, and for a few other references: the pieces' classes appear as two levels of the time; printed as a map in allocation order.
  python tools/class_map.py [N [PIECE_GB [REFS]]]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mono_dataset_code_amd import capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
PIECE = int(float(sys.argv[2]) * (1 << 30)) if len(sys.argv) > 2 else 1 << 30
REFS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 7, 50, 120]


def main():
    ctx = capi.Context(0)
    L, B = capi.hip_lib(), capi.bench_lib()
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    s = st.cuda_stream
    pieces = []
    for i in range(N):
        p = C.c_void_p()
        if L.mdc_device_alloc(ctx.handle, PIECE, C.byref(p)) != 0:
            break
        pieces.append(p.value)
    n = len(pieces)
    print("%d pieces of %.2f GB; virtual addresses %s" % (n, PIECE / 1e9, "descending" if pieces[1] < pieces[0] else "ascending"))

    def mix(rd, wr):
        ts = []
        for k in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert B.mdcb_ceiling_mix_device(0, rd, PIECE, wr, PIECE // 4, 16384, 0, s) == 0  # one dword written per 16 bytes read: 4 : 1 by bytes
            e1.record()
            torch.cuda.synchronize()
            if k:
                ts.append(e0.elapsed_time(e1))
        return min(ts)

    def mix_w(rd, wr):  # write-heavy: the path's own ratio, 1 byte read per 2 written
        ts = []
        for k in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert B.mdcb_ceiling_mix_device(0, rd, PIECE // 2, wr, PIECE, 16384, 0, s) == 0
            e1.record()
            torch.cuda.synchronize()
            if k:
                ts.append(e0.elapsed_time(e1))
        return min(ts)

    for ref in [r for r in REFS if r < n]:
        t = np.array([mix_w(pieces[ref], pieces[i]) if i != ref else np.nan for i in range(n)])
        lo, hi = np.nanmin(t), np.nanmax(t)
        thr = (lo + hi) / 2
        line = "".join("." if np.isnan(x) else ("S" if x > thr else "f") for x in t)
        print("read piece %3d, write piece i (1 : 2 bytes): %.4f .. %.4f ms; S = slow half, f = fast half, in allocation order:" % (ref, lo, hi))
        for k in range(0, n, 100):
            print("   %3d  %s" % (k, line[k:k + 100]))
        hist, edges = np.histogram(t[~np.isnan(t)], bins=8)
        print("   histogram of the times: " + "  ".join("%.3f:%d" % (edges[k], hist[k]) for k in range(8)))
        t2 = np.array([mix_w(pieces[i], pieces[ref]) if i != ref else np.nan for i in range(n)])
        lo2, hi2 = np.nanmin(t2), np.nanmax(t2)
        thr2 = (lo2 + hi2) / 2
        line2 = "".join("." if np.isnan(x) else ("S" if x > thr2 else "f") for x in t2)
        print("read piece i, write piece %3d: %.4f .. %.4f ms:" % (ref, lo2, hi2))
        for k in range(0, n, 100):
            print("   %3d  %s" % (k, line2[k:k + 100]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
