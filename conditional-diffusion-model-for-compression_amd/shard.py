"""Multi-GPU driver: the batch of images shards embarrassingly, one process per GPU,
no data-path collective.  Image i of a run is decoded by exactly one rank and keyed by its GLOBAL index
(noise streams, synthetic inputs), so the output is independent of the number of ranks.

torch.distributed (RCCL on GPUs, gloo in CPU tests) is used only for (a) the barrier that brackets a timed
region and (b) the max-over-ranks of the elapsed time / an optional gather of results outside the timed region.
"""
from __future__ import annotations

import time


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition of images [0, total) over `world` ranks: rank r gets [lo, hi)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def decode_shard(decode, total: int, rank: int, world: int, batch: int) -> dict:
    """Decode this rank's images in calls of at most `batch`; decode(first_image, count) -> sequence of `count`
    per-image results.  Returns {global index: result}."""
    lo, hi = shard_range(total, rank, world)
    out = {}
    for first in range(lo, hi, batch):
        n = min(batch, hi - first)
        res = decode(first, n)
        for k in range(n):
            out[first + k] = res[k]
    return out


def timed_region(fn, dist=None, sync=None) -> float:
    """barrier + sync, run fn, sync + barrier; returns the MAX over ranks of the elapsed seconds."""
    def fence():
        if sync is not None:
            sync()
        if dist is not None and dist.is_initialized():
            dist.barrier()
            if sync is not None:
                sync()
    fence()
    t0 = time.perf_counter()
    fn()
    if sync is not None:
        sync()
    elapsed = time.perf_counter() - t0
    fence()
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        import torch
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
