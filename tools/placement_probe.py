#!/usr/bin/env python3
"""Does the headline launch's speed depend on WHERE its buffers lie?

BENCH_r04 / r05: one process times the same kernel + plan at 1.57 ms on the headline's own 5.4 + 5.0 GB buffers and at 1.49 ms on the
first 4096 frames of the 50,000-frame sequence's 65 + 61 GB buffers; across "boxes" the headline varies by 5 %, the sequence by 0.3 %.
This probe times ONE plan on many placements inside one process:
  fresh   : separately allocated input / output pairs (all kept alive: distinct physical pages)
  arena   : one 48-GB allocation; input at a fixed offset, output at 8 GiB + delta for a list of deltas (and the mirror: output fixed,
            input moved)
  huge    : the first 4096 frames of a sequence-sized pair (65 + 61 GB), like bench.py's seq50k
Interleaved rounds, median of the launch times per placement.   python tools/placement_probe.py [rounds]
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from mono_dataset_code_amd import capi, synth  # noqa: E402

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
FRAMES, NPI, NPO = 4096, 1280 * 1024, 640 * 480
IN_BYTES, OUT_BYTES = FRAMES * NPI, FRAMES * NPO * 4
GiB = 1 << 30


def main():
    dev = torch.device("cuda", 0)
    d = synth.write_sequence_calibration(tempfile.mkdtemp(prefix="mdc_place_"))
    with bench.quiet_stdout():
        fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
        photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 1280, 1024)
    ctx = capi.Context(0)
    ctx.bind(fov, photo)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    s = st.cuda_stream
    flags = capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED | capi.RECTIFY
    info = ctx.info()
    rb, wb = int(info.src_bbox_bytes) * FRAMES // 16 * 16, OUT_BYTES

    places = []  # (name, in_ptr, out_ptr, keepalive)
    for k in range(6):
        a = torch.empty(IN_BYTES, dtype=torch.uint8, device=dev)
        b = torch.empty(OUT_BYTES, dtype=torch.uint8, device=dev)
        places.append(("fresh pair %d" % k, a.data_ptr(), b.data_ptr(), (a, b)))
    arena = torch.empty(24 * GiB, dtype=torch.uint8, device=dev)
    base = (arena.data_ptr() + (1 << 21) - 1) >> 21 << 21  # 2-MiB aligned
    for delta in (0, (1 << 20) + 4096):
        places.append(("arena: in +0, out +8 GiB + %d" % delta, base, base + 8 * GiB + delta, None))
    huge_in = torch.empty(50000 * NPI, dtype=torch.uint8, device=dev)
    huge_out = torch.empty(50000 * NPO * 4, dtype=torch.uint8, device=dev)
    places.append(("huge pair (65 + 61 GB), first 4096 frames", huge_in.data_ptr(), huge_out.data_ptr(), None))
    places.append(("huge pair, frames 20000..24095", huge_in.data_ptr() + 20000 * NPI, huge_out.data_ptr() + 20000 * NPO * 4, None))
    places.append(("huge input, fresh output 0", huge_in.data_ptr(), places[0][2], None))
    places.append(("fresh input 0, huge output", places[0][1], huge_out.data_ptr(), None))

    # ranges made with HIP's virtual memory management (hipMemCreate + hipMemMap; libmdc_bench: mdcb_chunked_alloc), one chunk and many
    import ctypes as C
    B = capi.bench_lib()

    def vmm(nbytes, chunk):
        n = (nbytes + chunk - 1) // chunk
        p = C.c_void_p()
        assert B.mdcb_chunked_alloc(0, chunk, n, 1, C.byref(p)) == 0
        return p.value

    # physically contiguous ranges (hipExtMallocWithFlags(hipDeviceMallocContiguous)): the largest page-table fragments there are
    path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
    hip = C.CDLL(path)
    hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]

    def contiguous(nbytes):
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, 0x4)
        return p.value if rc == 0 else None

    for k in range(int(os.environ.get("PLACE_CONTIG", "4"))):
        ci, co = contiguous(IN_BYTES), contiguous(OUT_BYTES)
        if ci and co:
            places.append(("contiguous pair %d" % k, ci, co, None))
            if k == 0:
                places.append(("contiguous input 0, fresh output 0", ci, places[0][2], None))
                places.append(("fresh input 0, contiguous output 0", places[0][1], co, None))
    v_in, v_out = vmm(IN_BYTES, 256 << 20), vmm(OUT_BYTES, 256 << 20)
    v_in1, v_out1 = vmm(IN_BYTES, 1 << 30), vmm(OUT_BYTES, 1 << 30)
    places.append(("VMM output (256-MiB chunks), fresh input 0", places[0][1], v_out, None))
    places.append(("VMM input + VMM output (256-MiB chunks)", v_in, v_out, None))
    places.append(("VMM input + VMM output (1-GiB chunks)", v_in1, v_out1, None))
    places.append(("VMM input, fresh output 0", v_in, places[0][2], None))
    seen = set()
    for name, pi, po, _ in places:
        if pi not in seen:
            ctx.synth_frames(pi, 0, FRAMES, NPI, synth.SEED, s)
            seen.add(pi)
    torch.cuda.synchronize()
    t = ctx.tune(places[0][1], places[0][2], FRAMES, flags, s)
    print("plan (tuned on fresh pair 0): tile %dx%d, %d frames per workgroup, %.4f ms; %s" % (t.tile_w, t.tile_h, t.frames_per_block, t.ms, ctx.describe_launch(flags, 0)))
    for _ in range(150):
        ctx.process_batch(places[0][1], places[0][2], FRAMES, flags, s)
    torch.cuda.synchronize()
    times = {p[0]: [] for p in places}
    ceil = {p[0]: [] for p in places}
    for r in range(ROUNDS):
        for name, pi, po, _ in places:
            for which, store in (("k", times), ("c", ceil)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                launch = (lambda: ctx.process_batch(pi, po, FRAMES, flags, s)) if which == "k" else (lambda: ctx.ceiling_mix(pi, rb, po, wb, 65536, 0, s))
                launch()
                e0.record()
                for _ in range(4):
                    launch()
                e1.record()
                torch.cuda.synchronize()
                store[name].append(e0.elapsed_time(e1) / 4)
    print("%-52s %10s %10s %8s | %12s" % ("placement", "median ms", "min ms", "frac8T", "ceiling ms"))
    alg = (int(info.src_bbox_bytes) + NPO * 4) * FRAMES
    for name, pi, po, _ in places:
        med = float(np.median(times[name]))
        print("%-52s %10.4f %10.4f %8.4f | %12.4f   in %%2MiB=%d out %%2MiB=%d" % (name, med, float(np.min(times[name])), alg / (med * 1e-3) / 8e12,
                                                                                 float(np.median(ceil[name])), pi % (1 << 21), po % (1 << 21)))
    meds = [float(np.median(times[n])) for n in times]
    print("spread of the medians: %.4f .. %.4f ms (%.1f %%)" % (min(meds), max(meds), 100 * (max(meds) / min(meds) - 1)))


if __name__ == "__main__":
    main()
