#!/usr/bin/env python3
"""Throughput of the photometric + FOV undistortion hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames B] [--workload fused|unmap|pyramid]

One "step" = one pass of the hot path over one batch of B synthetic 1280x1024 u8
frames that are already resident in HBM (one batched kernel launch through
mdc_process_batch_device).  Multi-GPU: one process per GPU (torchrun), the
sequence is sharded round-robin (frame f -> rank f % N), rank 0 builds the
calibration tables and broadcasts them once over RCCL; there is no collective
on the data path, so scaling is weak (B frames per GPU per step).

Rank 0 prints ONE JSON line: BASELINE.json's metric (Mpix/s of input pixels,
whole job), plus
  roofline     : algorithmic HBM bytes per launch / mean launch duration (HIP events
                 on the launching stream) against the 8 TB/s HBM peak
  cpu_baseline : the reference's own CPU path (oracle/_ref, all host cores) on a
                 bounded sample of the same frames, timed on this box (N=1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
IN_W, IN_H, OUT_W, OUT_H = 1280, 1024, 640, 480


def _flush_c_stdio():
    import ctypes

    ctypes.CDLL(None).fflush(None)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--preroll-s", type=float, default=0.3,
                   help="untimed clock ramp before the warmup steps: the first ~30 ms after an idle period run ~15 %% "
                        "slow (DVFS, profiles/r01_dvfs_warmup_curve.txt)")
    p.add_argument("--frames", type=int, default=1024, help="frames per GPU per step (batch of one launch)")
    p.add_argument("--workload", default="fused", choices=["fused", "unmap", "pyramid"])
    p.add_argument("--kernel", default="auto", choices=["auto", "gather", "tiled"])
    p.add_argument("--fpb", type=int, default=0, help="frames per workgroup (0 = library default)")
    p.add_argument("--tile-rows", type=int, default=0, help="output tile rows of the tiled kernel (0 = library default)")
    p.add_argument("--nbuf", type=int, default=0, help="LDS window buffers (0 = automatic)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    return p.parse_args()


def traffic_from_profiles(kernel_tag, frames_per_launch):
    """Measured HBM bytes per launch of the dominant kernel, from the separate PMC passes
    (tools/profile_bench.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, gfx950
    corrections applied) recorded per frame in profiles/hbm_traffic.json.  PMC cannot be
    collected inside this process, so the figure is the committed one; None if absent."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            e = json.load(f).get(kernel_tag)
        return round(e["bytes_per_frame"] * frames_per_launch) if e else None
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(args, calib_dir, flags_gvo):
    """The reference CPU path (unMapImage -> temp -> undistort<float>, as
    DatasetReader::getImage drives it) on all host cores, bounded sample."""
    from mono_dataset_code_amd import synth
    from oracle import loader

    cores = os.cpu_count() or 1
    npix = IN_W * IN_H
    nframes = 2 * cores
    frames = synth.noise_frames(0, nframes, npix)
    rect = args.workload != "unmap"
    if loader.have_ref() or os.path.isdir(loader.REFERENCE_ROOT):
        kind = "reference"
        R = loader.Ref()
        devnull = os.open(os.devnull, os.O_WRONLY)
        saved = os.dup(1)
        os.dup2(devnull, 1)  # the reference prints its calibration on construction
        try:
            fov = R.fov(os.path.join(calib_dir, "camera.txt"))
            photo = R.photo(os.path.join(calib_dir, "pcalib.txt"), os.path.join(calib_dir, "vignette.png"), IN_W, IN_H)
        finally:
            _flush_c_stdio()  # the chatter sits in libc's buffer; drop it into /dev/null, not after our JSON line
            os.dup2(saved, 1)
            os.close(devnull)

        def run(passes):
            return R.time_path(fov, photo, frames, cores, passes, rect, 1, 1, 1)
    else:
        kind = "port"
        cores = 1
        O = loader.Oracle()
        cam = O.parse_camera(os.path.join(calib_dir, "camera.txt"))
        t = O.fov_setup(cam)
        ginv, _ = O.photo_gamma(O.parse_pcalib(os.path.join(calib_dir, "pcalib.txt")))
        vinv = O.photo_vignette(synth.vignette_image(IN_W, IN_H))[1]

        def run(passes):
            return O.time_path(frames, passes, IN_W, IN_H, OUT_W, OUT_H, ginv, vinv, t["remap_x"], t["remap_y"], rect,
                               1, 1, 1)
    run(1)  # page in
    t1 = run(3) / 3
    passes = max(1, int(args.cpu_seconds / max(t1, 1e-3)))
    t = run(passes)
    fps = nframes * passes / t
    return {"value": round(fps * npix / 1e6, 1), "unit": "Mpix/s", "cores": cores, "kind": kind,
            "sample": "%d noise frames x %d passes, %d threads, %.1f s, flags g+v+o%s" %
                      (nframes, passes, cores, t, "+rectify" if rect else ""),
            "fps": round(fps, 1)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the hot path)")
    # one rank per GPU; MDC_BENCH_BACKEND=gloo lets the tests run several ranks on ONE GPU (RCCL refuses that)
    backend = os.environ.get("MDC_BENCH_BACKEND", "nccl")
    gpu = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mono_dataset_code_amd import capi, shard, synth

    # ---- calibration: rank 0 builds (host C++ classes), everyone imports the blob ----
    ctx = capi.Context(gpu)
    calib_dir = None
    blob = None
    if rank == 0:
        calib_dir = tempfile.mkdtemp(prefix="mdc_bench_")
        lines = synth.CAMERA_1280_TO_640 if args.workload != "pyramid" else synth.camera_lines(IN_W, IN_H, IN_W, IN_H)
        synth.write_sequence_calibration(calib_dir, lines)
        devnull = os.open(os.devnull, os.O_WRONLY)
        saved = os.dup(1)
        os.dup2(devnull, 1)  # the classes print the reference's calibration chatter on stdout
        try:
            fov = capi.UndistorterFOV(os.path.join(calib_dir, "camera.txt"))
            photo = capi.PhotometricUndistorter(os.path.join(calib_dir, "pcalib.txt"),
                                                os.path.join(calib_dir, "vignette.png"), IN_W, IN_H)
        finally:
            _flush_c_stdio()  # the chatter sits in libc's buffer; drop it into /dev/null, not after our JSON line
            os.dup2(saved, 1)
            os.close(devnull)
        assert fov.is_valid() and photo.valid() == 3
        blob = capi.pack_tables(fov, photo)  # host-side serialisation of GInv, vignetteInv, remapX/Y
    if world > 1:
        blob = shard.broadcast_tables(blob, src=0, device=coll_dev)  # the only collective: once, over RCCL
    ctx.import_tables(blob)  # every rank (rank 0 included) uploads the same bytes
    ctx.set_option(capi.OPT_KERNEL, {"auto": capi.KERNEL_AUTO, "gather": capi.KERNEL_GATHER, "tiled": capi.KERNEL_TILED}[args.kernel])
    ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, args.fpb)
    if args.tile_rows:
        ctx.set_option(capi.OPT_TILE_ROWS, args.tile_rows)
    if args.nbuf:
        ctx.set_option(capi.OPT_WINDOW_BUFFERS, args.nbuf)
    info = ctx.info()
    out_w, out_h = (info.out_w, info.out_h) if args.workload != "unmap" else (IN_W, IN_H)

    # ---- this rank's shard of the synthetic sequence, generated in HBM ----------------
    B = args.frames
    npix_in, npix_out = IN_W * IN_H, out_w * out_h
    tstream = torch.cuda.Stream(device=dev)  # every launch and every timing event goes on this stream
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    d_in = torch.empty(B * npix_in, dtype=torch.uint8, device=dev)
    d_out = torch.empty(B * npix_out, dtype=torch.float32, device=dev)
    for i, f in enumerate(shard.frames_of_rank(B * world, rank, world)):
        ctx.synth_frames(d_in.data_ptr() + i * npix_in, int(f), 1, npix_in, synth.SEED, stream)
    levels, d_levels = 4, []
    if args.workload == "pyramid":
        d_levels = [torch.empty(B * (out_w >> l) * (out_h >> l), dtype=torch.float32, device=dev) for l in range(1, levels)]
    flags = capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED | (0 if args.workload == "unmap" else capi.RECTIFY)

    def step():
        if args.workload == "pyramid":  # base + levels 1..3 in one launch
            ctx.process_pyramid_batch(d_in.data_ptr(), d_out.data_ptr(), levels, [t.data_ptr() for t in d_levels], B, flags, stream)
        else:
            ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), B, flags, stream)

    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.preroll_s:  # untimed: bring the clocks to their steady state
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps between barrier+sync on both sides -----------------
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        step()
        b.record()  # brackets the one kernel of a step (same stream as the launch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))

    if rank == 0:
        # spot parity of the benchmarked launch against the C oracle (2 frames)
        parity = None
        try:
            from oracle import loader
            O = loader.Oracle()
            rx, ry = fov.remap()
            _, vinv = photo.vignette()
            got = d_out[: 2 * npix_out].cpu().numpy().reshape(2, npix_out)
            raw = d_in[: 2 * npix_in].cpu().numpy().reshape(2, npix_in)
            bad = 0
            for f in range(2):
                want = O.get_image(raw[f], IN_W, IN_H, out_w, out_h, photo.ginv(), vinv, True, True, rx, ry,
                                   args.workload != "unmap", True, True, True)
                nw, ng = np.isnan(want), np.isnan(got[f])
                bad += int((nw != ng).sum()) + int((want[~nw & ~ng].view(np.uint32) != got[f][~nw & ~ng].view(np.uint32)).sum())
            parity = {"frames_checked": 2, "mismatching_pixels": bad}
        except OSError:
            pass

        frames_total = B * world * args.steps
        mpix = frames_total * npix_in / 1e6 / elapsed
        if args.workload == "unmap":
            alg_frame = npix_in * (1 + 4)
            tag = "unmap"
        else:
            alg_frame = int(info.src_bbox_bytes) + npix_out * 4
            tag = "fused_tiled" if info.tiled and args.kernel != "gather" else "fused_gather"
            if args.workload == "pyramid":  # + levels 1..3 written
                alg_frame += 4 * sum((out_w >> l) * (out_h >> l) for l in range(1, levels))
                tag = "pyramid_fused"
        achieved = alg_frame * B / (kernel_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic_from_profiles(tag, B),
                "kernel": tag, "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_frame": alg_frame,
                "frames_per_launch": B, "frac_of_measured_copy_ceiling_6290": round(achieved / 6290.0, 4)}
        out = {
            "metric": "Mpix/s photometric+FOV undistort, 1280x1024 gray",
            "value": round(mpix, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"fused": "configs[2]: fused photometric(g+v+o) + FOV bilinear remap 1280x1024 u8 -> 640x480 f32",
                                    "unmap": "configs[1]: unMapImage only (g+v+o) 1280x1024 u8 -> f32",
                                    "pyramid": "configs[4]: fused photometric + remap 1280x1024 -> 1280x1024 + 4-level box pyramid"}[args.workload],
                       "frames_per_gpu_per_step": B, "preroll_s": args.preroll_s, "sharding": "round-robin frame f -> rank f %% %d" % world,
                       "tables": "rank-0 build + one RCCL broadcast" if world > 1 else "local build",
                       "frames_per_s": round(frames_total / elapsed, 1),
                       "out_mpix_per_s": round(frames_total * npix_out / 1e6 / elapsed, 1)},
            "roofline": roof,
        }
        if parity is not None:
            out["parity"] = parity
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, calib_dir, flags)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
