#!/usr/bin/env python3
"""Throughput of the photometric + FOV undistortion hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames B]
                  [--workload fused|unmap|pyramid|seq50k]

One "step" = one pass of the hot path over one batch of B synthetic 1280x1024 u8
frames that are already resident in HBM (one batched kernel launch through
mdc_process_batch_device).  Multi-GPU: one process per GPU (torchrun), the
sequence is sharded round-robin (frame f -> rank f % N), rank 0 builds the
calibration tables and broadcasts them once over RCCL; there is no collective
on the data path.
  fused / unmap / pyramid : B frames per GPU per step                 -> weak scaling
  seq50k (BASELINE.json configs[3]) : ONE 50,000-frame sequence, rank r owns frames
          r, r+N, ...; a step = every rank's whole shard               -> strong scaling

Rank 0 prints ONE JSON line: BASELINE.json's metric (Mpix/s of input pixels,
whole job), plus
  roofline     : algorithmic HBM bytes per launch / launch duration (HIP events on the
                 launching stream; mean is the contract's figure, median and min beside
                 it) against the 8 TB/s HBM peak, AND against this box's own ceiling for
                 the same byte counts (a linear read+write stream timed in this process:
                 boxes differ by +-5 %, the ratio does not);
  cpu_baseline : the reference's own CPU path (oracle/_ref) on a bounded sample of the
                 same frames, timed on this box: all host cores (value), one thread (the
                 reference as shipped), and unMapImage alone (N=1 only).
"""
import argparse
import json
import os
import platform
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
IN_W, IN_H, OUT_W, OUT_H = 1280, 1024, 640, 480
SEQ50K = 50000


def _flush_c_stdio():
    import ctypes

    ctypes.CDLL(None).fflush(None)


class quiet_stdout:
    """The C++ classes (ours and the reference's) print their calibration on stdout; the contract wants ONE line."""

    def __enter__(self):
        sys.stdout.flush()
        self.devnull = os.open(os.devnull, os.O_WRONLY)
        self.saved = os.dup(1)
        os.dup2(self.devnull, 1)

    def __exit__(self, *a):
        _flush_c_stdio()  # the chatter sits in libc's buffer; drop it into /dev/null, not after our JSON line
        os.dup2(self.saved, 1)
        os.close(self.devnull)
        os.close(self.saved)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1500, help="timed steps (default: ~2.4 s of GPU time at 4096 frames per step)")
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--preroll-s", type=float, default=0.3,
                   help="untimed clock ramp before the warmup steps, at least this long: the first ~30 ms after an idle period "
                        "run ~15 %% slow (DVFS, profiles/r01_dvfs_warmup_curve.txt) and the memory system keeps speeding up "
                        "for seconds (profiles/r05_experiments/01_*).  The pre-roll is ADAPTIVE: it goes on until "
                        "--preroll-windows consecutive windows of --preroll-window-s agree within --preroll-tol, or --preroll-max-s")
    p.add_argument("--preroll-max-s", type=float, default=8.0)
    p.add_argument("--preroll-window-s", type=float, default=0.05)
    p.add_argument("--preroll-windows", type=int, default=5)
    p.add_argument("--preroll-tol", type=float, default=0.01)
    p.add_argument("--placement", default="auto", choices=["auto", "first", "malloc", "vmm"],
                   help="how the frame / result buffers of every workload are made: through the product's allocator, mdc_alloc_placed_device "
                        "(include/mdc_hip.h).  On MI355X the time of one launch depends on the allocations it runs on by up to 9 %% "
                        "(profiles/r05_experiments/05_*, 08_*, 10_*; DESIGN.md 6.1): auto = the library's default strategy, first = two plain "
                        "allocations as they come, malloc = candidates + every pair timed, vmm = ranges assembled from classified physical pieces")
    p.add_argument("--placement-candidates", type=int, default=0,
                   help="malloc strategy: candidate allocations per buffer (0 = the library's default, 6; 1 = same as --placement first)")
    p.add_argument("--no-again", action="store_true", help="do not time the headline a second time after the secondary workloads")
    p.add_argument("--parity-frames", type=int, default=16, help="frames of the benchmarked launch compared with the oracle")
    p.add_argument("--frames", type=int, default=0,
                   help="frames per GPU per step = batch of one launch (0 = 4096 for fused/unmap, 1024 for pyramid; "
                        "seq50k: the rank's shard)")
    p.add_argument("--workload", default="fused", choices=["fused", "unmap", "pyramid", "seq50k", "dso", "undistort_f32"])
    p.add_argument("--kernel", default="auto", choices=["auto", "gather", "tiled"])
    p.add_argument("--fpb", type=int, default=0, help="frames per workgroup (0 = library default)")
    p.add_argument("--tile-rows", type=int, default=0, help="output tile rows of the tiled kernel (0 = library default)")
    p.add_argument("--tile-cols", type=int, default=0, help="output tile columns of the tiled kernel (0 = library default)")
    p.add_argument("--nbuf", type=int, default=0, help="LDS window buffers (0 = automatic)")
    p.add_argument("--two-stage", type=int, default=0, choices=[0, 1, 2], help="wave-private strip kernel: 0 = library's choice, 1 = wherever plannable, 2 = off")
    p.add_argument("--no-tune", action="store_true", help="keep the library's built-in tile shape / frames per workgroup instead "
                                                          "of mdc_tune_device's measured choice (untimed, before the warm-up)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-secondary", action="store_true",
                   help="only the headline workload (default: after it, the other BASELINE.json configs are timed for "
                        "--secondary-steps steps each and reported under \"secondary\" in the same line)")
    p.add_argument("--secondary-steps", type=int, default=20)
    p.add_argument("--no-ceiling", action="store_true", help="skip the same-box linear-mix ceiling")
    p.add_argument("--markers", action="store_true",
                   help="bracket the headline's timed region with two launches of a no-op kernel (mdcb_marker_kernel, libmdc_bench.so) outside the timed "
                        "wall-clock region, on the stream of the timed launches: tools/profile_round.py cuts rocprofv3's trace / counters to the launches between them")
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="target CPU time of the all-core baseline sample")
    p.add_argument("--dump-dir", default="", help="test hook: every rank writes its first --dump-frames outputs here")
    p.add_argument("--dump-frames", type=int, default=0)
    return p.parse_args()


def traffic_from_profiles(kernel_name, frames_per_launch, code_id=None, path=None, workload=None):
    """Measured HBM bytes per launch of the dominant kernel, from the separate PMC passes
    (tools/profile_round.py: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, gfx950 corrections
    applied), recorded per frame in profiles/hbm_traffic.json under the kernel instantiation they
    were measured on AND the identity of the build they were measured on (mdc_code_id(): a hash of
    the kernel sources and compile flags, baked into the library).  PMC cannot be collected inside
    this process, so the figure is the committed one -- and only if it belongs to the instantiation
    AND the build that ran here; a changed kernel under an unchanged name gets None and the reason.
    -> (bytes per launch or None, source or reason)"""
    path = path or os.path.join(ROOT, "profiles", "hbm_traffic.json")
    stale = None
    try:
        with open(path) as f:
            entries = json.load(f)
            # the workload's own entry first (the fused launch and the 50,000-frame sequence run the same instantiation)
            order = sorted(entries.items(), key=lambda kv: 0 if workload and kv[0].startswith(workload + ":") else 1)
            for _, e in order:
                if e.get("kernel") != kernel_name:
                    continue
                if code_id is not None and e.get("code_id") != code_id:
                    stale = "stale: %s was measured on build %s, this is build %s" % (e.get("source"), e.get("code_id") or "(unrecorded)", code_id)
                    continue
                return round(e["bytes_per_frame"] * frames_per_launch), e.get("source")
    except (OSError, ValueError, KeyError):
        pass
    return None, stale


def gpu_sysfs_dir(index=0):
    """sysfs directory of HIP device `index`, found by its PCI address (a container sees every card of the node under
    /sys/class/drm, in another order than HIP's ordinals: card0 is usually somebody else's GPU)."""
    import glob

    try:
        p = torch.cuda.get_device_properties(index)
        d = "/sys/bus/pci/devices/%04x:%02x:%02x.0" % (int(getattr(p, "pci_domain_id", 0)), int(p.pci_bus_id), int(getattr(p, "pci_device_id", 0)))
        if os.path.exists(os.path.join(d, "pp_dpm_sclk")):
            return d
    except (AttributeError, RuntimeError, AssertionError, ValueError):
        pass
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
    return os.path.dirname(cards[index]) if len(cards) == 1 and index == 0 else None  # never guess among several cards


def gpu_clock_snapshot(index=0):
    """Current clocks / power / temperature of HIP device `index` from sysfs (amdgpu: the starred level of pp_dpm_*, hwmon),
    best effort: {} where the files are not there.  Cheap enough to call between timed regions."""
    import glob

    d = gpu_sysfs_dir(index)
    if d is None:
        return {}
    out = {"pci": os.path.basename(d)}
    for name in ("sclk", "mclk", "fclk", "socclk"):
        try:
            for line in open(os.path.join(d, "pp_dpm_" + name)):
                if line.rstrip().endswith("*"):
                    out[name + "_mhz"] = int(line.split(":")[1].strip().lower().split("mhz")[0])
        except (OSError, ValueError, IndexError):
            pass
    for name in ("current_memory_partition", "current_compute_partition", "mem_info_vram_total", "mem_info_vram_vendor"):  # NPS1 / SPX ...: how HBM is interleaved
        try:
            out[name.replace("current_", "").replace("mem_info_", "")] = open(os.path.join(d, name)).read().strip()
        except OSError:
            pass
    for name in ("gpu_busy_percent", "mem_busy_percent"):
        try:
            out[name] = int(open(os.path.join(d, name)).read())
        except (OSError, ValueError):
            pass
    for h in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
        for f, key, scale in (("power1_average", "power_w", 1e-6), ("power1_input", "power_w", 1e-6), ("temp1_input", "temp_c", 1e-3),
                              ("temp2_input", "temp2_c", 1e-3), ("temp3_input", "temp_mem_c", 1e-3)):
            try:
                out.setdefault(key, round(int(open(os.path.join(h, f)).read()) * scale, 1))
            except (OSError, ValueError):
                pass
    return out


def format_clock_snapshot(s):
    return " ".join("%s %s" % (k.replace("_mhz", "").replace("_percent", "%"), v) for k, v in s.items()) or "(no sysfs clocks)"


def usable_cpus():
    """Hardware threads, cut down to the container's CFS quota (cgroup v2 cpu.max / v1 cfs_quota_us): more busy
    threads than that are throttled, so this is the honest "cores" of the CPU baseline."""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(round(float(q) / float(p)))))
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, int(round(q / p))))
        except (OSError, ValueError):
            pass
    return n


def host_description():
    model = platform.processor() or "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        cc = subprocess.run(["g++", "--version"], stdout=subprocess.PIPE, text=True).stdout.splitlines()[0]
    except (OSError, IndexError):
        cc = "unknown"
    return model, cc


def cpu_baseline(args, calib_dir, rect):
    """The reference CPU path (unMapImage -> temp -> undistort<float>, as DatasetReader::getImage
    drives it, src/BenchmarkDatasetReader.h:222-223) on a bounded sample: all host cores (frames
    sharded over threads, objects shared) and one thread (the reference as shipped), plus the
    unMapImage-only analogue of configs[1]."""
    from mono_dataset_code_amd import synth
    from oracle import loader

    hw_threads = os.cpu_count() or 1
    cores = usable_cpus()  # = hardware threads unless the container has a CPU quota
    npix = IN_W * IN_H
    nframes = max(512, 2 * cores)  # 670 MB of raw frames: beyond the host's last-level cache (2 x 256 MB), as a sequence is
    frames = synth.noise_frames(0, nframes, npix)
    if loader.have_ref() or os.path.isdir(loader.REFERENCE_ROOT):
        kind = "reference"
        R = loader.Ref()
        with quiet_stdout():  # the reference prints its calibration on construction
            fov = R.fov(os.path.join(calib_dir, "camera.txt"))
            photo = R.photo(os.path.join(calib_dir, "pcalib.txt"), os.path.join(calib_dir, "vignette.png"), IN_W, IN_H)

        def run(nfr, threads, passes, rectify):
            return R.time_path(fov, photo, frames[:nfr], threads, passes, rectify, 1, 1, 1)
    else:
        kind = "port"
        cores = 1
        O = loader.Oracle()
        cam = O.parse_camera(os.path.join(calib_dir, "camera.txt"))
        t = O.fov_setup(cam)
        ginv, _ = O.photo_gamma(O.parse_pcalib(os.path.join(calib_dir, "pcalib.txt")))
        vinv = O.photo_vignette(synth.vignette_image(IN_W, IN_H))[1]

        def run(nfr, threads, passes, rectify):
            return O.time_path(frames[:nfr], passes, IN_W, IN_H, OUT_W, OUT_H, ginv, vinv, t["remap_x"], t["remap_y"],
                               rectify, 1, 1, 1)

    def rate(nfr, threads, rectify, seconds):
        run(nfr, threads, 1, rectify)  # page in
        t1 = run(nfr, threads, 2, rectify) / 2
        passes = max(1, int(seconds / max(t1, 1e-3)))
        t = run(nfr, threads, passes, rectify)
        fps = nfr * passes / t
        return {"value": round(fps * npix / 1e6, 1), "fps": round(fps, 1), "threads": threads,
                "sample": "%d noise frames x %d passes, %.1f s" % (nfr, passes, t)}

    allc = rate(nframes, cores, rect, args.cpu_seconds)
    model, cc = host_description()
    if kind == "reference" and hw_threads > cores:  # also with every hardware thread busy (the quota throttles them): report the better
        over = rate(nframes, hw_threads, rect, 3.0)
        over["note"] = "%d threads on %d hardware threads under a %d-CPU cgroup quota" % (hw_threads, hw_threads, cores)
        if over["value"] > allc["value"]:
            allc, over = over, allc
            cores_used = hw_threads
        else:
            cores_used = cores
    else:
        over, cores_used = None, cores
    out = {"value": allc["value"], "unit": "Mpix/s", "cores": cores_used, "kind": kind,
           "sample": "%s, %d threads, flags g+v+o%s" % (allc["sample"], allc["threads"], "+rectify" if rect else ""),
           "fps": allc["fps"], "cpu_model": model, "compiler": cc + " -O3 -DNDEBUG -std=c++0x (no -march, no FMA)",
           "hardware_threads": hw_threads, "cpu_quota": cores if cores < hw_threads else None}
    if over is not None:
        out["other_thread_count"] = {"value": over["value"], "fps": over["fps"], "threads": over["threads"], "sample": over["sample"]}
    if kind == "reference":
        one = rate(16, 1, rect, 3.0)
        out["one_thread_as_shipped"] = {"value": one["value"], "unit": "Mpix/s", "fps": one["fps"], "sample": one["sample"]}
        if rect:  # configs[1]'s CPU analogue: unMapImage only, full-size float output
            ua = rate(nframes, allc["threads"], False, 3.0)
            u1 = rate(16, 1, False, 2.0)
            out["unmap_only"] = {"all_cores": {"value": ua["value"], "fps": ua["fps"], "sample": ua["sample"]},
                                 "one_thread": {"value": u1["value"], "fps": u1["fps"], "sample": u1["sample"]}, "unit": "Mpix/s"}
    return out


WORKLOAD_TEXT = {
    "fused": "configs[2]: fused photometric(g+v+o) + FOV bilinear remap 1280x1024 u8 -> 640x480 f32",
    "unmap": "configs[1]: unMapImage only (g+v+o) 1280x1024 u8 -> f32",
    "pyramid": "configs[4]: fused photometric + remap 1280x1024 -> 1280x1024 + 4-level box pyramid",
    "seq50k": "configs[3]: one %d-frame sequence (fused photometric + remap -> 640x480), frame f on GPU f %% N",
    "undistort_f32": "SURVEY 8(a3): UndistorterFOV::undistort<float> on device-resident float frames (the output of unMapImage g+v+o, the reference's "
                     "two-call composition src/BenchmarkDatasetReader.h:222-223) 1280x1024 f32 -> 640x480 f32",
    "dso": "SURVEY 8(f4), beyond configs[4]: fused photometric + remap 1280x1024 -> 1280x1024 + box levels 1-3 + DSO's (I, dx, dy) and "
           "absSquaredGrad of every level -- NOT in the reference: definition and oracle are this repository's, parity unpinned",
}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks_if_needed(args):
    """`python bench.py --gpus N` without a launcher: become N ranks (torch.distributed.run, one per GPU), never a
    silent single rank.  Fewer than N visible devices is an error unless the collectives run over gloo
    (MDC_BENCH_BACKEND=gloo: the test mode in which several ranks share one GPU)."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, env_world))
        return
    if args.gpus <= 1:
        return
    backend = os.environ.get("MDC_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < args.gpus and backend != "gloo":
        raise SystemExit("bench.py: --gpus %d but %d GPU(s) visible; one rank per GPU over RCCL needs %d devices "
                         "(MDC_BENCH_BACKEND=gloo shares devices between ranks, for tests only)" % (args.gpus, ndev, args.gpus))
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the hot path)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


class Dist:
    """The process group of a run (or its absence): one rank per GPU."""

    def __init__(self, args):
        import torch.distributed as dist

        self.dist = dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # one rank per GPU; MDC_BENCH_BACKEND=gloo lets the tests run several ranks on ONE GPU (RCCL refuses that);
        # MDC_BENCH_FORCE_DIST=1 makes a single rank go through the process group too (RCCL init, broadcast, barrier,
        # all-reduce of a world of one) so that the collective branch is executable on a 1-GPU box.
        self.backend = os.environ.get("MDC_BENCH_BACKEND", "nccl")
        self.active = self.world > 1 or os.environ.get("MDC_BENCH_FORCE_DIST") == "1"
        ndev = torch.cuda.device_count()
        if self.backend == "nccl" and self.world > 1 and self.local_rank >= ndev:
            raise SystemExit("bench.py: rank %d (local %d) has no GPU of its own: %d visible" % (self.rank, self.local_rank, ndev))
        self.gpu = self.local_rank % ndev
        torch.cuda.set_device(self.gpu)
        self.dev = torch.device("cuda", self.gpu)
        self.coll_dev = self.dev if self.backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live
        if self.active:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)
            assert dist.get_world_size() == self.world, "process group has %d ranks, WORLD_SIZE says %d" % (dist.get_world_size(), self.world)

    def stream(self):
        if getattr(self, "_stream", None) is None:
            self._stream = torch.cuda.Stream(device=self.dev)
        return self._stream

    def barrier(self):
        if self.active:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if not self.active:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.coll_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, values):
        """Every rank's list of floats, in rank order."""
        if not self.active:
            return [list(values)]
        got = [torch.zeros(len(values), dtype=torch.float64, device=self.coll_dev) for _ in range(self.world)]
        self.dist.all_gather(got, torch.tensor(list(values), dtype=torch.float64, device=self.coll_dev))
        return [[float(x) for x in t.cpu()] for t in got]

    def devices(self):
        """Per rank: the device ordinal it runs on and that device's PCI bus id -- N ranks must show N different ones."""
        p = torch.cuda.get_device_properties(self.gpu)
        bus = float(getattr(p, "pci_bus_id", -1))
        dom = float(getattr(p, "pci_domain_id", 0))
        dev = float(getattr(p, "pci_device_id", 0))
        return [{"rank": r, "device": int(v[0]), "pci": "%04x:%02x:%02x" % (int(v[1]), int(v[2]), int(v[3]))}
                for r, v in enumerate(self.gather([float(self.gpu), dom, bus, dev]))]


def bits_differ(want, have):
    nw, ng = np.isnan(want), np.isnan(have)
    return int((nw != ng).sum()) + int((want[~nw & ~ng].view(np.uint32) != have[~nw & ~ng].view(np.uint32)).sum())


def spot_frames(B, fpb, n):
    """Local frame indices of a launch to compare with the oracle: the first and last frame of the first frame groups (a workgroup
    walks `fpb` consecutive frames), of a group in the middle, and of the tapered tail (groups of fpb/2, fpb/4, fpb/8 at the end of
    a large launch) -- the places where a frame-loop prologue, a hand-counted wait or the tail arithmetic would go wrong."""
    fpb = max(int(fpb or 1), 1)
    mid = (B // 2) // fpb * fpb
    cand = [0, B - 1, fpb - 1, fpb, B - 2, mid, mid - 1, mid + fpb - 1, B - max(fpb // 8, 1), B - max(fpb // 8, 1) - 1,
            B - max(fpb // 4, 1), B - max(fpb // 2, 1), B - fpb, B - fpb - 1, 2 * fpb - 1, 2 * fpb, 1, B - 3, mid + 1, 3 * fpb]
    if B <= n:
        return list(range(B))
    out = []
    for f in cand + [int(round(k * (B - 1) / max(n - 1, 1))) for k in range(n)]:  # (small launches: the candidates collapse; fill up evenly)
        if 0 <= f < B and f not in out:
            out.append(f)
        if len(out) >= n:
            break
    return sorted(out)


class Buf:
    """A device buffer of the bench with byte-offset reads: either a torch allocation (hipMalloc through PyTorch's allocator: levels, gradient
    images) or one half of a pair made by the product's allocator (mdc_alloc_placed_device: frames / base results)."""

    def __init__(self, nbytes, dev=None, ptr=None, ctx=None):
        self.nbytes = int(nbytes)
        self.ctx = ctx
        self.t = None
        if ptr is None:
            self.t = torch.empty(self.nbytes, dtype=torch.uint8, device=dev)
            self.ptr = self.t.data_ptr()
        else:
            self.ptr = int(ptr)

    def data_ptr(self):
        return self.ptr

    def read(self, offset_bytes, count, dtype):
        """count elements of dtype from byte offset, as a numpy array"""
        nb = count * np.dtype(dtype).itemsize
        if self.t is not None:
            return self.t[offset_bytes:offset_bytes + nb].cpu().numpy().view(dtype)
        torch.cuda.synchronize()
        return self.ctx.copy_to_host(self.ptr + offset_bytes, count, dtype)

    def free(self):
        self.t = None


class Workload:
    """One BASELINE.json config set up on every rank: context + tables, this rank's shard of the synthetic sequence in HBM, the
    tuned plan, `step()` = one pass of the hot path over the shard.  Kept alive so that the SAME launch (same context, same
    buffers, same plan) can be timed again later in the process."""

    def __init__(self, args, D, wl, frames, tune=True):
        from mono_dataset_code_amd import capi, shard, synth

        self.args, self.D, self.wl = args, D, wl
        world, rank, dev = D.world, D.rank, D.dev
        # ---- calibration: rank 0 builds (host C++ classes), everyone imports the blob ----
        self.rect = rect = wl != "unmap"
        self.ctx = ctx = capi.Context(D.gpu)
        self.calib_dir = self.fov = self.photo = blob = None
        if rank == 0:
            self.calib_dir = tempfile.mkdtemp(prefix="mdc_bench_")
            lines = synth.CAMERA_1280_TO_640 if wl not in ("pyramid", "dso") else synth.camera_lines(IN_W, IN_H, IN_W, IN_H)
            synth.write_sequence_calibration(self.calib_dir, lines)
            with quiet_stdout():
                self.fov = capi.UndistorterFOV(os.path.join(self.calib_dir, "camera.txt"))
                self.photo = capi.PhotometricUndistorter(os.path.join(self.calib_dir, "pcalib.txt"),
                                                         os.path.join(self.calib_dir, "vignette.png"), IN_W, IN_H)
            assert self.fov.is_valid() and self.photo.valid() == 3
            blob = capi.pack_tables(self.fov, self.photo)  # host-side serialisation of GInv, vignetteInv, remapX/Y
        self.bcast_ms = None
        if D.active:
            if D.backend == "nccl":
                torch.cuda.synchronize()
            t_b = time.perf_counter()
            sent = None if blob is None else blob.copy()
            blob = shard.broadcast_tables(blob, src=0, device=D.coll_dev, even_alone=True)  # the only collective: once, over RCCL
            if D.backend == "nccl":
                torch.cuda.synchronize()
            self.bcast_ms = (time.perf_counter() - t_b) * 1e3
            if sent is not None:
                assert np.array_equal(sent, blob), "the broadcast changed the root's own blob"
        self.blob_bytes = int(blob.size)
        ctx.import_tables(blob)  # every rank (rank 0 included) uploads the same bytes
        ctx.set_option(capi.OPT_KERNEL, {"auto": capi.KERNEL_AUTO, "gather": capi.KERNEL_GATHER, "tiled": capi.KERNEL_TILED}[args.kernel])
        ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, args.fpb)
        if args.tile_cols:
            ctx.set_option(capi.OPT_TILE_COLS, args.tile_cols)
        if args.tile_rows:
            ctx.set_option(capi.OPT_TILE_ROWS, args.tile_rows)
        if args.nbuf:
            ctx.set_option(capi.OPT_WINDOW_BUFFERS, args.nbuf)
        if args.two_stage:
            ctx.set_option(capi.OPT_TWO_STAGE, args.two_stage)
        self.info = info = ctx.info()
        self.out_w, self.out_h = (info.out_w, info.out_h) if rect else (IN_W, IN_H)

        # ---- this rank's shard of the synthetic sequence, generated in HBM ----------------
        if wl == "seq50k":
            total = frames * world if frames else SEQ50K  # --frames shrinks the sequence for tests
        else:
            # pyramid: 1024 frames = 1.3 GB in + 7.1 GB out.  Not fewer: below ~384 frames the raw batch (1.3 MB a frame) partly
            # survives in the 256-MiB Infinity Cache from one step to the next and the rate comes out up to 40 % too high
            # (tools/footprint_curve.py, profiles/r02c_footprint_curve.txt)
            # dso: 512 frames = 18.1 GB of outputs
            total = (frames or (1024 if wl in ("pyramid", "undistort_f32") else 512 if wl == "dso" else 4096)) * world
        self.total = total
        self.mine = mine = shard.frames_of_rank(total, rank, world)
        self.B = B = len(mine)
        self.npix_in, self.npix_out = IN_W * IN_H, self.out_w * self.out_h
        # every launch and every timing event goes on ONE stream, made once per process and shared by the workloads that follow each
        # other in it (a caller has one stream; and the chunked strip path of configs[4] alternates between the caller's stream and
        # one of its own -- on a fresh stream per workload the pair sometimes landed on one hardware queue and ran 18 % slower)
        tstream = D.stream()
        torch.cuda.set_stream(tstream)
        self.stream = stream = tstream.cuda_stream
        # Buffers: the frames and the (base) results come from the PRODUCT's allocator (mdc_alloc_placed_device), which picks -- by timing the
        # pass itself -- a pair of buffers the pass runs fast on; the bench has no search of its own.  What a caller gets who takes the first
        # allocations as they come is the allocator's ms_first, printed beside (roofline.placement).
        self.flags = capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED | (capi.RECTIFY if rect else 0)
        strategy = {"auto": capi.PLACE_AUTO, "first": capi.PLACE_FIRST, "malloc": capi.PLACE_MALLOC, "vmm": capi.PLACE_VMM}[args.placement]
        if args.placement_candidates == 1 or (D.active and D.backend != "nccl"):
            strategy = capi.PLACE_FIRST  # (ranks sharing one GPU over gloo, the test mode: buffers as they come)
        elif args.placement_candidates > 1:
            os.environ["MDC_PLACE_CANDIDATES"] = str(args.placement_candidates)
        torch.cuda.synchronize()
        self.f32 = wl == "undistort_f32"  # the frames are floats (4 bytes a pixel): a larger frame buffer, placed by the same probe
        self.placed = ctx.alloc_placed(B, self.flags, strategy, stream, in_bytes=B * self.npix_in * 4 if self.f32 else 0)
        self.d_in = Buf(self.placed.in_bytes, ptr=self.placed.d_in, ctx=ctx)
        self.d_out = Buf(self.placed.out_bytes, ptr=self.placed.d_out, ctx=ctx)
        self.placement = self.placed.describe()
        self.d_raw = None
        raw_ptr = self.d_in.data_ptr()
        if self.f32:  # the float frames are what unMapImage makes of the synthetic raw frames (kept: the parity check starts from them)
            self.d_raw = Buf(B * self.npix_in, dev)
            raw_ptr = self.d_raw.data_ptr()
        if world == 1:
            ctx.synth_frames(raw_ptr, 0, B, self.npix_in, synth.SEED, stream)
        else:
            for i, f in enumerate(mine):  # local frame i = global frame rank + i * world
                ctx.synth_frames(raw_ptr + i * self.npix_in, int(f), 1, self.npix_in, synth.SEED, stream)
        if self.f32:
            ctx.unmap_batch(raw_ptr, self.d_in.data_ptr(), B, self.flags & ~capi.RECTIFY, stream)
        self.levels, self.d_levels = 4, []
        self.d_dI, self.d_abs = [], []
        # the step's further outputs (config 5's levels, the DSO hand-off's gradient images): from the product's allocator too where the pair
        # was assembled (mdc_alloc_striped_set_device: every buffer striped over the device's memory classes), else plain allocations
        counts = []
        if wl in ("pyramid", "dso"):
            counts += [B * (self.out_w >> l) * (self.out_h >> l) for l in range(1, self.levels)]
        if wl == "dso":  # per level: (I, dx, dy) triples + absSquaredGrad
            counts += [B * (self.out_w >> l) * (self.out_h >> l) * 3 for l in range(self.levels)] + [B * (self.out_w >> l) * (self.out_h >> l) for l in range(self.levels)]
        self.extra = None
        if counts and self.placed.strategy == capi.PLACE_VMM:
            self.extra = ctx.alloc_striped_set([4 * n_ for n_ in counts], stream)
            bufs = [Buf(4 * n_, ptr=self.extra.d_ptr[k], ctx=ctx) for k, n_ in enumerate(counts)]
            self.placement["further_outputs"] = self.extra.note.decode(errors="replace")
        else:
            bufs = [Buf(4 * n_, dev) for n_ in counts]
        if wl in ("pyramid", "dso"):
            self.d_levels = bufs[:self.levels - 1]
        if wl == "dso":
            self.d_dI, self.d_abs = bufs[self.levels - 1:2 * self.levels - 1], bufs[2 * self.levels - 1:]
        if os.environ.get("MDC_BENCH_DEBUG_BUFFERS"):  # where every buffer lies (a GPU memory fault names an address)
            for name, b in [("in", self.d_in), ("out", self.d_out)] + [("level%d" % (i + 1), b) for i, b in enumerate(self.d_levels)] + \
                    [("dI%d" % i, b) for i, b in enumerate(self.d_dI)] + [("abs%d" % i, b) for i, b in enumerate(self.d_abs)]:
                sys.stderr.write("buffer %-7s 0x%x .. 0x%x (%d bytes, %s)\n" % (name, b.data_ptr(), b.data_ptr() + b.nbytes, b.nbytes, "torch.empty"))
        self.tuned = None
        if tune and wl in ("fused", "seq50k") and args.kernel == "auto" and not (args.no_tune or args.fpb or args.tile_rows or args.tile_cols or args.nbuf):
            # plan selection by measurement, on (a part of) this rank's own batch; untimed set-up like the table build
            t = ctx.tune(self.d_in.data_ptr(), self.d_out.data_ptr(), min(B, 4096), self.flags, stream)
            self.tuned = {"tile": [t.tile_w, t.tile_h], "frames_per_workgroup": t.frames_per_block, "candidates": t.candidates,
                          "ms_on_%d_frames" % min(B, 4096): round(t.ms, 4)}
            self.info = ctx.info()
        self.kernel_name = ctx.describe_launch(self.flags, self.levels if wl in ("pyramid", "dso") else -1 if self.f32 else 0)
        if wl == "dso":
            self.kernel_name += " + gradients_levels_kernel"
        if wl == "unmap":
            self.alg_read, self.alg_write = self.npix_in, self.npix_in * 4
        else:
            self.alg_read, self.alg_write = int(self.info.src_bbox_bytes) * (4 if self.f32 else 1), self.npix_out * 4
            if wl in ("pyramid", "dso"):  # + levels 1..3 written
                self.alg_write += 4 * sum((self.out_w >> l) * (self.out_h >> l) for l in range(1, self.levels))
            if wl == "dso":  # + 3 + 1 floats per pixel of every level (the levels' re-read by the gradient launch is not algorithmic)
                self.alg_write += 16 * sum((self.out_w >> l) * (self.out_h >> l) for l in range(self.levels))
        self.alg_frame = self.alg_read + self.alg_write

    def step(self):
        if self.wl == "dso":  # base + levels + gradient images in one call (chunks chosen by the library)
            self.ctx.process_pyramid_gradients_batch(self.d_in.data_ptr(), self.d_out.data_ptr(), self.levels, [t.data_ptr() for t in self.d_levels],
                                                     [t.data_ptr() for t in self.d_dI], [t.data_ptr() for t in self.d_abs], self.B, self.flags, 0, self.stream)
        elif self.f32:
            self.ctx.undistort_batch_f32(self.d_in.data_ptr(), self.d_out.data_ptr(), self.B, self.stream)
        elif self.wl == "pyramid":  # base + levels 1..3 in one call
            self.ctx.process_pyramid_batch(self.d_in.data_ptr(), self.d_out.data_ptr(), self.levels, [t.data_ptr() for t in self.d_levels],
                                           self.B, self.flags, self.stream)
        else:
            self.ctx.process_batch(self.d_in.data_ptr(), self.d_out.data_ptr(), self.B, self.flags, self.stream)

    def free(self):
        for b in self.d_levels + self.d_dI + self.d_abs + [self.d_raw]:
            if b is not None:
                b.free()
        self.d_in = self.d_out = self.d_raw = None
        self.d_levels, self.d_dI, self.d_abs = [], [], []
        torch.cuda.synchronize()
        if self.extra is not None:
            self.ctx.free_striped_set(self.extra)
            self.extra = None
        if self.placed is not None:
            self.ctx.free_placed(self.placed)
            self.placed = None
        self.ctx.close()
        torch.cuda.empty_cache()

    def preroll(self, min_s=None):
        """Untimed: launches until the step time stands still.  Windows of ~preroll_window_s of back-to-back steps (one event
        pair each); done when the last `preroll_windows` windows agree within `preroll_tol` (and at least min_s have passed),
        or after preroll_max_s.  A fixed 0.3 s was not a steady state on every box (BENCH_r04: the same launch 1.61 ms in the
        headline, 1.49 ms fifteen seconds later)."""
        a = self.args
        min_s = a.preroll_s if min_s is None else min_s
        need = max(2, a.preroll_windows)
        t0 = time.perf_counter()
        wins, n = [], 3
        while True:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                self.step()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            wins.append(ms)
            n = max(2, min(4096, int(a.preroll_window_s * 1e3 / max(ms, 1e-3)) + 1))
            el = time.perf_counter() - t0
            last = wins[-need:]
            converged = len(wins) > need and (max(last) - min(last)) <= a.preroll_tol * min(last)
            if (converged and el >= min_s) or el >= max(a.preroll_max_s, min_s):
                break
        return {"seconds": round(el, 3), "windows": len(wins), "converged": bool(converged), "first_window_ms": round(wins[0], 4),
                "slowest_window_ms": round(max(wins), 4), "last_window_ms": round(wins[-1], 4),
                "rule": "%d consecutive windows of %.0f ms within %.1f %%, at least %.1f s, at most %.1f s" %
                        (need, a.preroll_window_s * 1e3, a.preroll_tol * 100, min_s, max(a.preroll_max_s, min_s))}

    def timed(self, steps, warmup):
        """W untimed warm-up steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; max over ranks."""
        D = self.D
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        if self.args.markers:
            self.ctx.marker(1, self.stream)
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, b in evs:
            a.record()
            self.step()
            b.record()  # brackets the launches of a step (same stream as the launch)
        clocks = gpu_clock_snapshot(D.gpu)  # host-side file reads while the queued steps run: the clocks UNDER LOAD
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        elapsed = D.max_over_ranks(time.perf_counter() - t0)
        if self.args.markers:
            self.ctx.marker(2, self.stream)
            torch.cuda.synchronize()
        ktimes = np.array([a.elapsed_time(b) for a, b in evs], dtype=np.float64)
        kstat = [float(ktimes.mean()), float(np.median(ktimes)), float(ktimes.min())]
        per_rank = [[round(x, 4) for x in k] for k in D.gather(kstat)]
        return {"elapsed": elapsed, "steps": steps, "kstat": kstat, "per_rank_kernel_ms": per_rank, "clocks": clocks}

    def dump(self):
        """test hook: every rank hands out its first outputs (checked against the oracle per GLOBAL frame index)"""
        args, rank = self.args, self.D.rank
        if not (args.dump_dir and args.dump_frames > 0):
            return
        n = min(args.dump_frames, self.B)
        np.save(os.path.join(args.dump_dir, "rank%d_out.npy" % rank), self.d_out.read(0, n * self.npix_out, np.float32).reshape(n, self.npix_out))
        np.save(os.path.join(args.dump_dir, "rank%d_idx.npy" % rank), np.asarray(self.mine[:n], dtype=np.int64))
        np.save(os.path.join(args.dump_dir, "rank%d_in_head.npy" % rank), np.stack([self.d_in.read(i * self.npix_in, 64, np.uint8) for i in range(n)]))

    def ceiling(self):
        """Same-box yardstick: the traffic mix of this launch as a linear stream, no arithmetic (rank 0)."""
        ctx, B, stream = self.ctx, self.B, self.stream
        wbytes = min(self.alg_write * B, self.d_out.nbytes)
        rbytes = min(self.alg_read * B, self.d_in.nbytes) // 16 * 16
        reps = 12
        best = None
        for blocks, span in ((4096, 0), (16384, 0), (65536, 0), (4096, 1), (16384, 1), (65536, 1)):
            cev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for _ in range(3):
                ctx.ceiling_mix(self.d_in.data_ptr(), rbytes, self.d_out.data_ptr(), wbytes, blocks, span, stream)
            for a, b in cev:
                a.record()
                ctx.ceiling_mix(self.d_in.data_ptr(), rbytes, self.d_out.data_ptr(), wbytes, blocks, span, stream)
                b.record()
            torch.cuda.synchronize()
            c = np.array([a.elapsed_time(b) for a, b in cev])
            if best is None or np.median(c) < np.median(best[0]):
                best = (c, blocks, span)
        cms, cblocks, cspan = best
        # the ceiling kernel wrote over the outputs: redo the step so that parity checks real results
        self.step()
        torch.cuda.synchronize()
        scale = (self.alg_read * B + self.alg_write * B) / float(rbytes + wbytes)  # pyramid: levels are separate buffers, the stream writes d_out only
        return {"ms_median": round(float(np.median(cms)) * scale, 4), "ms_min": round(float(cms.min()) * scale, 4),
                "read_bytes": self.alg_read * B, "write_bytes": self.alg_write * B,
                "what": "linear 16-B reads + wave-contiguous nt dword writes of the launch's ALGORITHMIC bytes, same process; "
                        "fastest of 6 launch shapes: %d workgroups, %s" % (cblocks, "contiguous span per workgroup" if cspan else "grid-stride")}

    def parity(self, nframes):
        """Spot parity of the benchmarked launch's outputs against the C oracle (rank 0): `nframes` frames spread over the launch
        (spot_frames), every pyramid level for configs[4]."""
        from mono_dataset_code_amd import synth

        try:
            from oracle import loader
            O = loader.Oracle()
        except OSError as e:  # no oracle library on this box: say so, never drop the field
            return "unavailable (%s)" % str(e)[:120]
        rx, ry = self.fov.remap()
        _, vinv = self.photo.vignette()
        ginv = self.photo.ginv()
        # the base image against the REFERENCE ITSELF where its build travelled with the tree (oracle/_ref/libmdc_ref.so: the
        # reference's own sources, own tables from the same calibration files), else against the C restatement on our tables
        R = rfov = rphoto = None
        if loader.have_ref():
            R = loader.Ref()
            with quiet_stdout():
                rfov = R.fov(os.path.join(self.calib_dir, "camera.txt"))
                rphoto = R.photo(os.path.join(self.calib_dir, "pcalib.txt"), os.path.join(self.calib_dir, "vignette.png"), IN_W, IN_H)
        bad = nan_bad = black_bad = 0
        max_rel = 0.0
        fpb = int(getattr(self.info, "frames_per_block", 0) or (self.tuned or {}).get("frames_per_workgroup", 0) or 64)
        checked = spot_frames(self.B, fpb, max(2, nframes))
        npi, npo = self.npix_in, self.npix_out
        for f in checked:
            raw = (self.d_raw if self.f32 else self.d_in).read(f * npi, npi, np.uint8)
            assert np.array_equal(raw, synth.noise_frames(int(self.mine[f]), 1, npi)[0]), "frame %d is not global frame %d" % (f, self.mine[f])
            if R is not None:
                want = R.get_image(rfov, rphoto, raw, self.rect, True, True, True)
            else:
                want = O.get_image(raw, IN_W, IN_H, self.out_w, self.out_h, ginv, vinv, True, True, rx, ry, self.rect, True, True, True)
            got = self.d_out.read(f * npo * 4, npo, np.float32)
            bad += bits_differ(want, got)
            # SURVEY.md 8(d)'s report for the base image, beside the bit count: NaN mask, exact zeros (black), relative error elsewhere
            nan_bad += int(np.count_nonzero(np.isnan(want) != np.isnan(got)))
            black_bad += int(np.count_nonzero((want == 0) != (got == 0)))
            fin = np.isfinite(want) & (want != 0) & np.isfinite(got)
            if fin.any():
                max_rel = max(max_rel, float(np.max(np.abs((got[fin].astype(np.float64) - want[fin]) / want[fin]))))
            src, cw, ch = want, self.out_w, self.out_h
            for l in range(self.levels if self.d_levels else 1):
                if l > 0:
                    src = O.pyramid_level(src, cw, ch)
                    cw, ch = cw // 2, ch // 2
                    bad += bits_differ(src, self.d_levels[l - 1].read(f * cw * ch * 4, cw * ch, np.float32))
                if self.d_dI:
                    w_dI, w_abs = O.gradients(src, cw, ch)
                    bad += bits_differ(w_dI.reshape(-1), self.d_dI[l].read(f * cw * ch * 12, cw * ch * 3, np.float32))
                    bad += bits_differ(w_abs, self.d_abs[l].read(f * cw * ch * 4, cw * ch, np.float32))
        out = {"frames_checked": len(checked), "frames": checked, "levels_checked": 1 + len(self.d_levels), "mismatching_pixels": bad,
               "nan_mask_mismatches": nan_bad, "black_pixel_mismatches": black_bad, "max_relative_error": max_rel,
               "against": "oracle/_ref/libmdc_ref.so (the reference's own sources compiled here, its own tables from the same calibration files)"
                          if R is not None else "oracle/liboracle.so (C restatement, pinned to the reference build by tests/test_oracle_vs_ref.py)"}
        if self.d_levels:
            out["pinned"] = "base image: by the reference; box levels%s: parity unpinned (not in the reference, own definition and oracle)" % (
                " and gradient images" if self.d_dI else "")
        return out

    def frac_of(self, kernel_ms):
        return self.alg_frame * self.B / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS

    def roofline(self, timing, ceiling):
        from mono_dataset_code_amd import capi

        info, B, D = self.info, self.B, self.D
        kernel_ms, kernel_med, kernel_min = timing["kstat"]
        achieved = self.alg_frame * B / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = traffic_from_profiles(self.kernel_name, B, capi.code_id(), workload=self.wl)
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": self.kernel_name, "kernel_ms": round(kernel_ms, 4), "kernel_ms_median": round(kernel_med, 4),
                "kernel_ms_min": round(kernel_min, 4), "frac_at_median": round(self.frac_of(kernel_med), 4),
                "algorithmic_bytes_per_frame": self.alg_frame, "algorithmic_read_bytes_per_frame": self.alg_read,
                "frames_per_launch": B, "frac_of_measured_copy_ceiling_6290": round(achieved / 6290.0, 4),
                "tile": ([info.f32_tile_w, info.f32_tile_h] if info.f32_tiled else None) if self.f32 else [info.tile_w, info.tile_h] if self.rect and info.tiled else None, "window_buffers": info.window_buffers if self.rect else None,
                "clocks_in_timed_region": timing["clocks"] or None}
        if self.wl == "dso":
            roof["launches_per_step"] = "per chunk of frames: one remap launch (base + levels 1-3) + one gradients_levels_kernel launch over all levels"
        if self.wl == "pyramid" and info.prefetch_chunk and B >= 2 * info.prefetch_chunk:
            # the strip path walks the batch in chunks: per chunk one linear prefetch of the next chunk's source rows into the
            # Infinity Cache + one remap launch.  kernel_ms is the time of ALL launches of a step (HIP events around the
            # call); the prefetch's reads are extra traffic, not algorithmic bytes.
            nchunk = -(-B // info.prefetch_chunk)
            roof["launches_per_step"] = {"remap_strip_kernel": nchunk, "prefetch_rows_kernel": nchunk, "frames_per_chunk": info.prefetch_chunk,
                                         "streams": info.prefetch_streams}
        if ceiling is not None:
            roof["same_box_mix_ceiling"] = ceiling
            roof["frac_of_same_box_mix_ceiling"] = round(ceiling["ms_median"] / kernel_med, 4)
        if D.active:
            roof["per_rank_kernel_ms_mean_median_min"] = timing["per_rank_kernel_ms"]
            roof["per_rank_frac"] = [round(self.frac_of(k[0]), 4) for k in timing["per_rank_kernel_ms"]]
        return roof

    def result(self, timing, ceiling, parity, preroll):
        D, total, steps, elapsed = self.D, self.total, timing["steps"], timing["elapsed"]
        frames_total = total * steps
        text = WORKLOAD_TEXT[self.wl] % total if self.wl == "seq50k" else WORKLOAD_TEXT[self.wl]
        return {
            "value": round(frames_total * self.npix_in / 1e6 / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 4), "steps": steps,
            "timed_region_ms": round(elapsed * 1e3, 3),  # what `value` rests on: steps x ms_per_step of GPU time (after the adaptive pre-roll)
            "scaling": "strong" if self.wl == "seq50k" else "weak", "parity": parity, "roofline": self.roofline(timing, ceiling),
            "config": {"workload": text, "frames_per_gpu_per_step": self.B, "sequence_frames": total if self.wl == "seq50k" else None,
                       "preroll": preroll, "sharding": "round-robin frame f -> rank f %% %d" % D.world,
                       "tables": ("rank-0 build + one %s broadcast" % ("RCCL" if D.backend == "nccl" else D.backend)) if D.active else "local build",
                       "collective_backend": D.backend if D.active else None,
                       "table_broadcast_ms": round(self.bcast_ms, 3) if self.bcast_ms is not None else None,
                       "table_blob_bytes": self.blob_bytes,
                       "plan": self.tuned if self.tuned is not None else "built-in",
                       "placement": self.placement,
                       "frames_per_s": round(frames_total / elapsed, 1),
                       "out_mpix_per_s": round(frames_total * self.npix_out / 1e6 / elapsed, 1)},
        }


def measure(W, steps, warmup, do_ceiling, parity_frames, min_preroll=None, dump=False):
    """pre-roll to a steady state, time, (rank 0:) yardstick + spot parity -> the workload's result dictionary on rank 0, None elsewhere"""
    pre = W.preroll(min_preroll)
    timing = W.timed(steps, warmup)
    if dump:
        W.dump()
    if W.D.rank != 0:
        return None
    ceiling = W.ceiling() if do_ceiling else None
    return W.result(timing, ceiling, W.parity(parity_frames), pre)


def main():
    args = parse()
    launch_ranks_if_needed(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the hot path)")
    D = Dist(args)
    from mono_dataset_code_amd import capi

    wl = args.workload
    do_ceiling = not args.no_ceiling
    clocks_idle = gpu_clock_snapshot(D.gpu)
    H = Workload(args, D, wl, args.frames)
    head = measure(H, args.steps, args.warmup, do_ceiling, args.parity_frames, dump=True)
    # every rank's buffers come from the product's allocator; its probe times (median of 5 launches of the pass over min(B, 4096) frames, on the
    # built-in plan) on the pair it handed out and on the first two plain allocations of the device, per rank
    place_all = D.gather([float(H.placed.ms_first), float(H.placed.ms_chosen), float(H.placed.strategy)])
    if D.rank == 0:
        pf, pc = float(H.placed.ms_first), float(H.placed.ms_chosen)
        head["roofline"]["placement"] = {
            "buffers": "made by mdc_alloc_placed_device (libmdc_hip.so; config.placement says how): the time of one launch depends on the allocations it "
                       "runs on, 1.48 to 1.61 ms in one process (profiles/r05_experiments/05_*, 08_*, 10_*; DESIGN.md 6.1)",
            "probe_ms_on_first_allocations": round(pf, 4) if pf else None, "probe_ms_on_chosen_pair": round(pc, 4) if pc else None,
            "frac_on_first_allocation": round(head["roofline"]["frac"] * pc / pf, 4) if pf and pc else None,
            "frac_on_first_allocation_is": "frac x (probe on the chosen pair / probe on the first allocations): what a caller gets who takes two hipMalloc's as they come"}
    devices = D.devices() if D.active else [{"rank": 0, "device": D.gpu}]
    for d_, pl in zip(devices, place_all):
        d_["placement"] = {"strategy": capi.PLACE_NAMES.get(int(pl[2]), "?"), "probe_ms_on_first_allocations": round(pl[0], 4) or None,
                           "probe_ms_on_chosen_pair": round(pl[1], 4) or None}
    # ---- the other BASELINE.json configs, timed in the same process (same box, same clocks) -------------------------
    # N = 1: configs[1] unMapImage, configs[4] pyramid, configs[3] as one 50,000-frame sequence on the one GPU.
    # N > 1: configs[3] as BASELINE.json words it -- the one sequence dealt round-robin over the ranks (strong scaling).
    secondary = None
    if wl == "fused" and not args.no_secondary and not args.frames:
        secondary = {}
        todo = ("unmap", "undistort_f32", "pyramid", "dso", "seq50k") if D.world == 1 else ("seq50k",)
        for w2 in todo:
            torch.cuda.empty_cache()
            W2 = Workload(args, D, w2, {"seq50k": 0, "dso": 512}.get(w2, 1024))
            r = measure(W2, min(args.steps, args.secondary_steps), 10, do_ceiling, min(args.parity_frames, 4))
            W2.free()
            if r is not None:
                rf = r["roofline"]
                secondary[w2] = {"workload": r["config"]["workload"], "value": r["value"], "unit": "Mpix/s", "scaling": r["scaling"],
                                 "steps": r["steps"], "ms_per_step": r["ms_per_step"], "frames_per_gpu_per_step": r["config"]["frames_per_gpu_per_step"],
                                 "kernel": rf["kernel"], "kernel_ms": rf["kernel_ms"], "frac": rf["frac"],
                                 "frac_of_same_box_mix_ceiling": rf.get("frac_of_same_box_mix_ceiling"),
                                 "algorithmic_bytes_per_frame": rf["algorithmic_bytes_per_frame"],
                                 "traffic": rf["traffic"], "traffic_source": rf["traffic_source"],
                                 "launches_per_step": rf.get("launches_per_step"), "per_rank_frac": rf.get("per_rank_frac"),
                                 "plan": r["config"]["plan"], "placement": r["config"]["placement"], "preroll": r["config"]["preroll"], "parity": r["parity"]}
    # ---- the headline AGAIN: same context, same buffers, same plan, after everything else ran ------------------------------
    # (the same launch was timed 8 % apart within one process; both figures are printed, they must agree)
    again = None
    if secondary is not None and not args.no_again:
        pre2 = H.preroll()
        t2 = H.timed(args.steps, args.warmup)
        if D.rank == 0:
            c2 = H.ceiling() if do_ceiling else None
            k2 = t2["kstat"]
            again = {"value": round(H.total * t2["steps"] * H.npix_in / 1e6 / t2["elapsed"], 1), "ms_per_step": round(t2["elapsed"] / t2["steps"] * 1e3, 4),
                     "kernel_ms": round(k2[0], 4), "kernel_ms_median": round(k2[1], 4), "frac": round(H.frac_of(k2[0]), 4),
                     "frac_of_same_box_mix_ceiling": round(c2["ms_median"] / k2[1], 4) if c2 else None,
                     "same_box_mix_ceiling_ms": c2["ms_median"] if c2 else None, "preroll": pre2, "clocks_in_timed_region": t2["clocks"] or None,
                     "seconds_after_first": None}
    if D.rank == 0:
        out = {
            "metric": "Mpix/s photometric+FOV undistort, 1280x1024 gray",
            "value": head["value"], "unit": "Mpix/s", "n_gpus": D.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "timed_region_ms": head["timed_region_ms"], "higher_is_better": True, "scaling": head["scaling"],
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": head["config"], "roofline": head["roofline"], "parity": head["parity"],
            "ranks": {"world": D.world, "backend": D.backend if D.active else None,
                      "rccl_ranks": D.dist.get_world_size() if D.active and D.backend == "nccl" else None, "devices": devices},
            "build_flags": capi.build_flags(), "code_id": capi.code_id(), "clocks_idle_at_start": clocks_idle or None,
        }
        if again is not None:
            again.pop("seconds_after_first")
            out["again"] = again
            out["roofline"]["frac_again"] = again["frac"]
            out["roofline"]["frac_again_over_frac"] = round(again["frac"] / max(out["roofline"]["frac"], 1e-9), 4)
        if secondary is not None:
            out["secondary"] = secondary
        if D.world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, H.calib_dir, H.rect)
        print(json.dumps(out), flush=True)
    if D.active:
        D.dist.barrier()
        D.dist.destroy_process_group()


if __name__ == "__main__":
    main()
