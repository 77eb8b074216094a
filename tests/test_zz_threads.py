"""Threading contract of the boundary ("Threading"): the reference's unMapImage and undistort are
re-entrant on a shared object (read-only tables, caller-owned buffers; undistort is const), and 8 threads sharing one
pair of objects is how a multi-threaded caller would drive them.  Here: 8 host threads share ONE PhotometricUndistorter,
ONE UndistorterFOV and ONE mdc_ctx and call the class methods and the C-ABI host / device entry points concurrently
(ctypes releases the GIL for the duration of a call), every thread on its own frames; every result must be the
oracle's, bit for bit.

The file is named test_zz_* on purpose: the driver runs `pytest -x`, and stress / concurrency tests go LAST in collection
order so that a flake here can never keep the parity tests from running.

Stream discipline (round-2 post-mortem): every torch operation that touches a buffer a kernel on `stream` also touches
runs under `with torch.cuda.stream(stream)` -- torch.cuda.Stream() is a hipStreamNonBlocking stream, so work enqueued on
torch's default (null) stream is NOT ordered with it; round 2's version poisoned d_out with fill_ on the null stream
and the fill could land after the kernel had written its results.  tests/native/thread_soak.cpp is the same contract
without torch (hipMemsetAsync / launch / hipMemcpyAsync all on the thread's own stream)."""
import itertools
import os
import subprocess
import threading

import numpy as np
import pytest

from conftest import bits_equal
from test_gpu_parity import Setup

pytestmark = pytest.mark.gpu

NTHREADS, ROUNDS = 8, 12


def test_shared_objects_from_many_threads(calib_dirs, oracle):
    import torch

    from mono_dataset_code_amd import capi, synth

    s = Setup("small_explicit", calib_dirs, oracle)
    npi, npo = s.W * s.H, s.w * s.h
    all_flags = [(capi.RECTIFY * r) | (capi.GAMMA * g) | (capi.VIGNETTE * v) | (capi.KILL_OVEREXPOSED * o)
                 for r, g, v, o in itertools.product((0, 1), repeat=4)]
    rng = np.random.default_rng(5)
    frames = [rng.integers(0, 256, npi, dtype=np.uint8) for _ in range(NTHREADS)]
    for f in frames:
        f[rng.integers(0, npi, 50)] = 255
    # expected results, computed up front on one thread
    want_proc = [{fl: s.want(oracle, f, bool(fl & capi.RECTIFY), bool(fl & capi.GAMMA), bool(fl & capi.VIGNETTE),
                             bool(fl & capi.KILL_OVEREXPOSED)) for fl in all_flags} for f in frames]
    want_unmap = [oracle.unmap(f, s.ginv, s.vinv, True, True, 1, 1, 1) for f in frames]
    want_und = [oracle.undistort(u, s.rx, s.ry, s.W) for u in want_unmap]
    want_und8 = [oracle.undistort(f, s.rx, s.ry, s.W) for f in frames]
    errors = []
    start = threading.Barrier(NTHREADS)

    def worker(k):
        try:
            raw = frames[k]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                d_in = torch.from_numpy(np.stack([raw] * 3)).cuda()
                d_out = torch.empty((3, npo), dtype=torch.float32, device="cuda")
            stream.synchronize()
            start.wait()
            for it in range(ROUNDS):
                # the two class methods, as DatasetReader::getImage chains them (reference src/BenchmarkDatasetReader.h:222-223)
                tmp = np.full(npi, -7.0, np.float32)
                s.photo.unmap(raw, tmp, 1, 1, 1)
                assert bits_equal(tmp, want_unmap[k]), ("unMapImage", k, it)
                out = np.full(npo, -7.0, np.float32)
                s.fov.undistort(tmp, out)
                assert bits_equal(out, want_und[k]), ("undistort<float>", k, it)
                out8 = np.full(npo, -7.0, np.float32)
                s.fov.undistort(raw, out8)
                assert bits_equal(out8, want_und8[k]), ("undistort<uchar>", k, it)
                # the fused host call with a flag combination that differs per thread and iteration
                fl = all_flags[(k * 5 + it) % 16]
                o2 = np.full(npo if fl & capi.RECTIFY else npi, -7.0, np.float32)
                s.ctx.process_host(raw, o2, fl)
                assert bits_equal(o2, want_proc[k][fl]), ("process_host", k, it, fl)
                # a device batch on the thread's own stream
                with torch.cuda.stream(stream):  # poison, launch and read-back are all ordered on `stream`
                    d_out.fill_(-7.0)
                    s.ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), 3, 15, stream.cuda_stream)
                    got = d_out.cpu().numpy()
                for j in range(3):
                    assert bits_equal(got[j], want_proc[k][15]), ("process_batch", k, it, j)
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(NTHREADS)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


SOAK = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "thread_soak")


@pytest.mark.parametrize("name,threads,iters", [("small_explicit", 8, 400), ("full_1280_to_640", 8, 24)])
def test_native_threads_one_context_own_streams(name, threads, iters, calib_dirs):
    """The same contract without torch (tests/native/thread_soak.cpp): T std::threads, ONE mdc_ctx, every thread its own
    hipStreamNonBlocking stream with hipMemsetAsync poison -> mdc_process_batch_device -> hipMemcpyAsync on THAT stream,
    flags rotating, plus the blocking host entry points, all against the C oracle.  (tools/soak_threads.sh runs it for
    2000+ iterations per thread; this is the short version.)"""
    if not os.path.exists(SOAK):
        pytest.skip("thread_soak not built")
    r = subprocess.run([SOAK, calib_dirs[name], str(threads), str(iters)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("THREAD_SOAK")]
    assert r.returncode == 0 and line, r.stdout[-3000:]
    kv = dict(zip(line[-1].split()[1::2], line[-1].split()[2::2]))
    assert int(kv["mismatches"]) == 0 and int(kv["call_failures"]) == 0 and int(kv["device_launches"]) == threads * iters, line[-1]


@pytest.mark.parametrize("name,threads,iters,batch,chunk", [("mag4_full_black", 8, 150, 5, 1), ("upsample", 12, 60, 7, 2)])
def test_native_threads_share_the_second_stream_pool(name, threads, iters, batch, chunk, calib_dirs):
    """The chunked strip path borrows its second stream from the context's slot pool (at most 8 slots, shared with the host
    entry points): 8 / 12 threads launching batches that go in 1- / 2-frame chunks over two streams, on ONE context, next to
    blocking host calls -- every result against the C oracle; a thread that finds no free slot runs its chunks on one stream."""
    if not os.path.exists(SOAK):
        pytest.skip("thread_soak not built")
    r = subprocess.run([SOAK, calib_dirs[name], str(threads), str(iters), str(batch), str(chunk)], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("THREAD_SOAK")]
    assert r.returncode == 0 and line, r.stdout[-3000:]
    kv = dict(zip(line[-1].split()[1::2], line[-1].split()[2::2]))
    assert int(kv["mismatches"]) == 0 and int(kv["call_failures"]) == 0 and int(kv["device_launches"]) == threads * iters, line[-1]
    assert int(kv["prefetch_chunk"]) == chunk
