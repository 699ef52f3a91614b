"""The N > 1 path on CPU: two gloo ranks shard a batch, decode with the (CPU) oracle standing in for the device
decode, and the gathered result equals the single-process decode bit for bit.  No GPU needed."""
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


def _decode_fn():
    import oracle
    cfg = cdx.unet_config(**TINY)
    params = cdx.init_params(cfg, seed=SEED)

    def decode(first, count):
        out = []
        for i in range(first, first + count):           # one image per call: identical arithmetic in every layout
            cond = torch.from_numpy(cdx.synthetic_batch(cfg, SEED, i, 1)["cond"])
            out.append(oracle.sample_ref(cfg, params, cond, STEPS, seed=SEED, method="ddpm", first_image=i)[0].numpy())
        return out
    return decode


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        decode = _decode_fn()
        res = {}
        elapsed = cdx.timed_region(lambda: res.update(cdx.decode_shard(decode, TOTAL, rank, world, batch=2)), dist)
        gathered = [None] * world
        dist.all_gather_object(gathered, res)
        if rank == 0:
            merged = {}
            for g in gathered:
                assert not (set(g) & set(merged))          # every image decoded by exactly one rank
                merged.update(g)
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
    single = cdx.decode_shard(_decode_fn(), TOTAL, 0, 1, batch=TOTAL)
    assert sorted(merged) == list(range(TOTAL)) and elapsed > 0
    for i in range(TOTAL):
        assert np.array_equal(merged[i], single[i]), i
