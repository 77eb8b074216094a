#!/usr/bin/env python3
"""Write (and read) bandwidth of a LINEAR stream per device allocation.

tools/placement_probe.py: one process, one box, the same kernel -- 1.48 ms on one pair of buffers, 1.61 ms on another; the linear-mix
ceiling kernel follows the OUTPUT buffer (1.20 .. 1.40 ms).  This probe allocates N buffers of SIZE_GB one after the other (hipMalloc
through mdc_device_alloc, no caching allocator), times a pure linear write of each (nontemporal dword stores, 65536 workgroups) and a
pure linear read, frees every other one, allocates again and times again: is the rate a property of the allocation (where it lies,
how it is mapped), does it depend on the order, does memory that was used before come back the same?
  python tools/alloc_probe.py [N [SIZE_GB]]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mono_dataset_code_amd import capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SIZE = int(float(sys.argv[2]) * (1 << 30)) if len(sys.argv) > 2 else 5 << 30


def main():
    ctx = capi.Context(0)
    L = capi.hip_lib()
    B = capi.bench_lib()
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    s = st.cuda_stream
    scratch = C.c_void_p()
    assert L.mdc_device_alloc(ctx.handle, 1 << 20, C.byref(scratch)) == 0

    def alloc():
        p = C.c_void_p()
        rc = L.mdc_device_alloc(ctx.handle, SIZE, C.byref(p))
        return p.value if rc == 0 else None

    def rate(ptr, kind):
        ts = []
        for k in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if kind == "w":
                rc = B.mdcb_ceiling_mix_device(0, None, 0, ptr, SIZE, 65536, 0, s)
            else:  # read SIZE bytes, write a quarter of that (one dword per 16-byte chunk read: the kernel's loop is driven by its stores)
                rc = B.mdcb_ceiling_mix_device(0, ptr, SIZE, ptr, SIZE // 4, 65536, 0, s)
            assert rc == 0
            e1.record()
            torch.cuda.synchronize()
            if k:
                ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))

    def rate_bytes(ptr, nbytes):
        ts = []
        for k in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert B.mdcb_ceiling_mix_device(0, None, 0, ptr, nbytes, 65536, 0, s) == 0
            e1.record()
            torch.cuda.synchronize()
            if k:
                ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))

    bufs = []
    for i in range(N):
        p = alloc()
        if p is None:
            print("allocation %d failed" % i)
            break
        bufs.append(p)
    print("%d allocations of %.1f GB (hipMalloc), in order; linear write / read-dominated stream, ms and TB/s" % (len(bufs), SIZE / 1e9))
    first = []
    for i, p in enumerate(bufs):
        w, r = rate(p, "w"), rate(p, "r")
        first.append(w)
        print("alloc %2d  va 0x%012x (%%1GiB = %4d MiB)  write %.4f ms = %.2f TB/s   read(+1/4 write) %.4f ms = %.2f TB/s" %
              (i, p, (p % (1 << 30)) >> 20, w, SIZE / w / 1e9, r, SIZE * 1.25 / r / 1e9), flush=True)
    print("write: min %.4f max %.4f ms (%.1f %% spread)" % (min(first), max(first), 100 * (max(first) / min(first) - 1)))
    # free every other buffer, allocate the same number again: do the new ones behave like the ones that lay there?
    freed = []
    for i in range(0, len(bufs), 2):
        L.mdc_device_free(ctx.handle, C.c_void_p(bufs[i]))
        freed.append(i)
    print("freed %s, allocating again" % freed)
    for i in freed:
        p = alloc()
        bufs[i] = p
        w = rate(p, "w")
        print("re-alloc in place of %2d  va 0x%012x  write %.4f ms = %.2f TB/s   (was %.4f)" % (i, p, w, SIZE / w / 1e9, first[i]), flush=True)
    # physically contiguous allocations (hipExtMallocWithFlags(hipDeviceMallocContiguous)): is "fast" = "contiguous"?
    for i in range(len(bufs)):
        if bufs[i]:
            L.mdc_device_free(ctx.handle, C.c_void_p(bufs[i]))
    path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
    hip = C.CDLL(path)
    hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipFree.argtypes = [C.c_void_p]
    for flag, name in ((0x4, "contiguous"), (0x0, "default")):
        got = []
        for i in range(int(os.environ.get("ALLOC_CONTIG_N", "8"))):
            p = C.c_void_p()
            rc = hip.hipExtMallocWithFlags(C.byref(p), SIZE, flag)
            if rc != 0:
                print("hipExtMallocWithFlags(%s) %d failed: %d" % (name, i, rc))
                break
            got.append(p.value)
        ws = [rate(p, "w") for p in got]
        print("hipExtMallocWithFlags(%s): write ms per allocation: %s" % (name, " ".join("%.3f" % w for w in ws)))
        for p in got:
            hip.hipFree(C.c_void_p(p))
    # a range mapped from separately created physical chunks, in order and permuted (mdcb_chunked_alloc): does SPREADING a buffer's pieces
    # over the physical address space make the fast class?
    for chunk_mib in (2, 32, 256, 1024):
        cb = chunk_mib << 20
        n = SIZE // cb
        for stride in (1, 7, 61, 1021):
            if stride >= n and stride != 1:
                continue
            ws = []
            for rep in range(3):
                p = C.c_void_p()
                rc = B.mdcb_chunked_alloc(0, cb, n, stride, C.byref(p))
                if rc != 0:
                    ws.append(float("nan"))
                    break
                ws.append(n * cb / rate_bytes(p.value, n * cb) / 1e9)
                B.mdcb_alias_free(0, p, cb, n)
            print("chunked range: %4d-MiB chunks x %5d, mapped with stride %4d: write TB/s %s" % (chunk_mib, n, stride, " ".join("%.2f" % w for w in ws)), flush=True)
    # the same through torch's caching allocator
    t = [torch.empty(SIZE, dtype=torch.uint8, device="cuda") for _ in range(3)]
    for i, x in enumerate(t):
        w = rate(x.data_ptr(), "w")
        print("torch.empty %d  va 0x%012x  write %.4f ms = %.2f TB/s" % (i, x.data_ptr(), w, SIZE / w / 1e9))


if __name__ == "__main__":
    main()
