"""Frame -> rank assignment of a sequence.

Frames are independent (reference src/BenchmarkDatasetReader.h:188-243 reads only
immutable tables plus frame `id`), so a sequence shards round-robin with no
exchange step: frame f belongs to rank f % world.  The only collective of the
whole path is the one-time broadcast of the calibration-table blob from rank 0
(broadcast_tables), done with the caller's torch.distributed backend -- "nccl"
(= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import numpy as np


def frames_of_rank(n_frames, rank, world):
    """Global indices of the frames `rank` owns, ascending: rank, rank+world, ..."""
    return np.arange(rank, n_frames, world, dtype=np.int64)


def count_of_rank(n_frames, rank, world):
    return max(0, (n_frames - rank + world - 1) // world)


def owner_of(frame, world):
    return frame % world


def local_index(frame, world):
    return frame // world


def broadcast_tables(blob, src=0, device=None, even_alone=False):
    """Broadcast the table blob (numpy uint8, as mdc_export_tables wrote it) from
    rank `src` to every rank; returns the blob each rank should import.  Ranks
    other than src may pass None.  A world of one returns the blob untouched
    unless `even_alone` (bench.py under MDC_BENCH_FORCE_DIST=1: the collective then
    really runs -- size word, payload, device round trip -- so that every line of the
    multi-GPU path executes on a one-GPU box)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or (dist.get_world_size() == 1 and not even_alone):
        return blob
    dev = device if device is not None else "cpu"
    size = torch.zeros(1, dtype=torch.int64, device=dev)
    if dist.get_rank() == src:
        size[0] = int(blob.size)
    dist.broadcast(size, src=src)
    n = int(size.item())
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(blob)).to(dev)
    else:
        t = torch.empty(n, dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()
