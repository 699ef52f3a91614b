"""The N > 1 path on CPU: two gloo ranks run the SAME driver bench.py uses (cdx.shard.ShardJob) with the CPU oracle
injected as the sampler; the gathered result equals the single-process decode bit for bit -- for a plain batch and
for a tiled (cfg5-like) job.  No GPU needed."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cdx

TINY = dict(image_size=16, base_channels=32, channel_mult=(1, 2), attn_resolutions=(8,), num_res_blocks=1)
TOTAL, STEPS, SEED = 5, 2, 11


def test_shard_range_partitions_exactly():
    for total in (0, 1, 5, 16, 128, 131):
        for world in (1, 2, 3, 8):
            spans = [cdx.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert cdx.shard_range(128, 3, 8) == (48, 64)         # BASELINE.json configs[2]: 128 images -> 16 per GPU
    with pytest.raises(ValueError):
        cdx.shard_range(4, 2, 2)


class OracleSampler:
    """Injected in place of the HIP sampler (ShardJob make_sampler): the CPU oracle, one image per oracle call so the
    arithmetic is identical in every batch / shard layout."""

    def __init__(self, cfg, run, params, device):
        self.cfg, self.run, self.params = cfg, run, params

    def sample(self, cond, steps, *, seed=0, first_image=0):
        import oracle
        return torch.stack([oracle.sample_ref(self.cfg, self.params, cond[k:k + 1], steps, seed=seed, method=self.run["method"],
                                              first_image=first_image + k)[0] for k in range(cond.shape[0])])

    def sample_tiled(self, cond, steps, *, overlap, seed=0, first_image=0):
        import oracle
        return torch.cat([oracle.sample_tiled_ref(self.cfg, self.params, cond[k:k + 1], steps, overlap=overlap, seed=seed,
                                                  first_image=first_image + k) for k in range(cond.shape[0])])


PLAIN = (cdx.unet_config(**TINY), dict(batch=TOTAL, steps=STEPS, method="ddpm"))
TILED = (cdx.unet_config(**TINY), dict(batch=3, steps=1, method="ddim", image=32, overlap=0))     # 32^2 image = 2x2 tiles of 16^2


def _jobs(rank, world):
    """The SAME driver bench.py uses (cdx.shard.ShardJob / decode), with the oracle injected as the sampler."""
    return [cdx.shard.ShardJob(None, "test", rank=rank, world=world, seed=SEED, images_per_call=2, make_sampler=OracleSampler, config=c)
            for c in (PLAIN, TILED)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = [{}, {}]
        jobs = _jobs(rank, world)
        elapsed = cdx.timed_region(lambda: [r.update(j.decode()) for r, j in zip(res, jobs)], dist)
        gathered = [None] * world
        dist.all_gather_object(gathered, [{i: v.numpy() for i, v in r.items()} for r in res])
        if rank == 0:
            merged = [{}, {}]
            for g in gathered:
                for m, part in zip(merged, g):
                    assert not (set(part) & set(m))          # every image decoded by exactly one rank
                    m.update(part)
            q.put((elapsed, merged))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_decode_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    elapsed, merged = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    torch.set_num_threads(1)
    assert elapsed > 0
    for (cfg, run), got, job in zip((PLAIN, TILED), merged, _jobs(0, 1)):
        job.images_per_call = run["batch"]                      # single process, one call
        single = job.decode()
        assert sorted(got) == list(range(run["batch"]))
        side = run.get("image", cfg["image_size"])
        for i in range(run["batch"]):
            assert got[i].shape == (3, side, side)
            assert np.array_equal(got[i], single[i].numpy()), i


def test_named_jobs_express_baseline_configs():
    """configs[2] (128 images -> 16 per GPU) and configs[4] (64 x 1024^2 -> 8 per GPU x 25 tiles) through the driver's
    own arithmetic (no sampler is built: a stub is injected)."""
    stub = lambda cfg, run, params, device: None      # noqa: E731
    spans = [cdx.shard.ShardJob(None, "cfg3", rank=r, world=8, make_sampler=stub, params={}) for r in range(8)]
    assert [(j.lo, j.hi) for j in spans] == [(16 * r, 16 * r + 16) for r in range(8)]
    assert all(j.total == 128 and j.steps == 100 and j.images_per_call == 16 and not j.tiled for j in spans)
    j5 = cdx.shard.ShardJob(None, "cfg5", rank=3, world=8, make_sampler=stub, params={})
    assert (j5.lo, j5.hi, j5.tiled, j5.steps, j5.run["image"], j5.run["overlap"]) == (24, 32, True, 50, 1024, 64)
    assert len(cdx.tile_origins(1024, 256, 64)) ** 2 == 25
    assert tuple(j5.cond(24, 1).shape) == (1, 3, 64, 64)
