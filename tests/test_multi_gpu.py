"""Multi-GPU plumbing: round-robin frame sharding and the one-time calibration-table
broadcast.  The data path has no collective; the broadcast is the
caller's torch.distributed backend -- RCCL ("nccl") on GPUs, gloo here: world_size 2 on CPU.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_round_robin_sharding_covers_every_frame_once():
    from mono_dataset_code_amd import shard

    for n, world in ((50000, 8), (17, 4), (3, 8), (0, 2), (1024, 1)):
        seen = np.zeros(n, np.int32)
        for r in range(world):
            fr = shard.frames_of_rank(n, r, world)
            assert len(fr) == shard.count_of_rank(n, r, world)
            assert all(shard.owner_of(int(f), world) == r for f in fr)
            assert [shard.local_index(int(f), world) for f in fr] == list(range(len(fr)))
            seen[fr] += 1
        assert np.all(seen == 1)
    # config 4 of BASELINE.json: 50k frames over 8 GPUs -> 6250 each
    assert [shard.count_of_rank(50000, r, 8) for r in range(8)] == [6250] * 8


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from mono_dataset_code_amd import capi, shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
d = %(calib)r
blob = None
if rank == 0:   # only rank 0 touches the calibration files
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), *fov.dims()[:2])
    blob = capi.pack_tables(fov, photo)
got = shard.broadcast_tables(blob, src=0)
# every rank ends with the same bytes; rank 1 cross-checks against tables it builds itself
digest = torch.tensor([int(np.frombuffer(got.tobytes(), np.uint8).astype(np.uint64).sum()), got.size], dtype=torch.int64)
all_d = [torch.zeros_like(digest) for _ in range(world)]
dist.all_gather(all_d, digest)
assert all(torch.equal(all_d[0], x) for x in all_d), all_d
if rank == 1:
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), *fov.dims()[:2])
    assert np.array_equal(capi.pack_tables(fov, photo), got)
# the shard arithmetic bench.py uses
mine = shard.frames_of_rank(37, rank, world)
cnt = torch.tensor([len(mine)]); dist.all_reduce(cnt); assert int(cnt) == 37
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", got.size)
'''


def test_table_broadcast_gloo_world2(calib_dirs, tmp_path):
    script = tmp_path / "worker.py"
    port = 29500 + (os.getpid() % 500)
    script.write_text(WORKER % {"root": ROOT, "port": port, "calib": calib_dirs["small_crop"]})
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
        assert "rank %d ok" % r in o


@pytest.mark.gpu
def test_packed_blob_equals_exported_blob(calib_dirs):
    """Host-side pack (what rank 0 broadcasts) == device context export, and importing it
    reproduces the context."""
    from mono_dataset_code_amd import capi

    d = calib_dirs["small_explicit"]
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), *fov.dims()[:2])
    ctx = capi.Context(0)
    ctx.bind(fov, photo)
    packed = capi.pack_tables(fov, photo)
    assert np.array_equal(packed, ctx.export_tables())
    c2 = capi.Context(0)
    c2.import_tables(packed)
    assert np.array_equal(c2.export_tables(), packed)
    i1, i2 = ctx.info(), c2.info()
    assert (i1.in_w, i1.out_w, i1.tiled, i1.n_tiles, i1.src_bbox_bytes) == (i2.in_w, i2.out_w, i2.tiled, i2.n_tiles, i2.src_bbox_bytes)


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu(tmp_path):
    """bench.py's N>1 path end to end (rank-0 table build, broadcast, import on every rank, round-robin
    shard, barrier + max-over-ranks timing, one JSON line from rank 0) with two ranks sharing the one GPU
    of the test box: gloo carries the collectives there (RCCL wants one device per rank)."""
    import json

    env = dict(os.environ, MDC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--frames", "64", "--preroll-s", "0.05"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]  # gloo announces itself on stdout, ranks interleaved
    assert len(lines) == 1, r.stdout  # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 5
    assert out["config"]["frames_per_gpu_per_step"] == 64 and "cpu_baseline" not in out
    assert out["parity"]["mismatching_pixels"] == 0
    assert out["config"]["tables"].startswith("rank-0 build")
