"""Multi-GPU driver: the batch of images shards embarrassingly, one process per GPU,
no data-path collective.  Image i of a run is decoded by exactly one rank and keyed by its GLOBAL index
(noise streams, synthetic inputs), so the output is independent of the number of ranks.

ONE driver serves bench.py (--gpus N), the CPU gloo test (tests/test_shard.py, with an injected sampler) and any
caller that wants "decode these N images of config X on this node":

    job = ShardJob(total_images, "cfg3", rank=r, world=n)     # BASELINE.json configs[2]: 128 images -> 16 per GPU
    images = job.decode()                                      # {global index: [C,H,W] tensor}
    decode(total_images, cfg_name, rank=..., world=...)        # the same, functional form

torch.distributed (RCCL on GPUs, gloo in CPU tests) is used only for (a) the barrier that brackets a timed
region and (b) the max-over-ranks of the elapsed time / an optional gather of results outside the timed region.
Processes must be started by torch.distributed.run (or mp.spawn) BEFORE anything touches the GPU: this module
never re-executes or forks.
"""
from __future__ import annotations

import time

# images per sampler call and GPU when the caller does not say (BASELINE.json configs: 16 / 16 / 8 / 8 per GPU)
IMAGES_PER_CALL = {"cfg1": 1, "cfg2": 16, "cfg3": 16, "cfg4": 8, "cfg5": 8}


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


def _hip_sampler(cfg: dict, run: dict, params: dict, device, **unet_kw):
    """The product sampler: HIP UNet + Sampler on `device` (raises without a GPU / libcdx.so: no CPU fallback)."""
    from .sampler import Sampler
    from .unet import UNet
    return Sampler(UNet(cfg, params, device=device, **unet_kw), method=run["method"])


class ShardJob:
    """This rank's part of a decode job: images [lo, hi) of `total_images` of a named configuration.

    make_sampler(cfg, run, params, device) -> object with .sample(cond, steps, seed=, first_image=) (and
    .sample_tiled(cond, steps, overlap=, seed=, first_image=) for tiled configs).  Default: the HIP sampler.
    config=(unet_cfg, run_cfg) replaces the named configuration (tests use a tiny network through the same driver).
    """

    def __init__(self, total_images: int | None, cfg_name: str = "cfg2", *, rank: int = 0, world: int = 1, device=None,
                 seed: int = 0, images_per_call: int | None = None, steps: int | None = None, make_sampler=None,
                 config: tuple[dict, dict] | None = None, params: dict | None = None, unet_kw: dict | None = None):
        from .config import named_config
        self.cfg, self.run = config if config is not None else named_config(cfg_name)
        self.cfg_name, self.rank, self.world, self.seed = cfg_name, rank, world, seed
        self.total = self.run["batch"] if total_images is None else total_images
        self.lo, self.hi = shard_range(self.total, rank, world)
        self.steps = self.run["steps"] if steps is None else steps
        self.images_per_call = images_per_call or IMAGES_PER_CALL.get(cfg_name, 1)
        self.tiled = "image" in self.run
        self.device = device
        self._params = params
        needs = getattr(make_sampler, "needs_params", True)      # a stand-in that ignores the weights need not have them drawn
        self.sampler = (make_sampler(self.cfg, self.run, self.params if needs else None, device) if make_sampler is not None else
                        _hip_sampler(self.cfg, self.run, self.params, device, **(unet_kw or {})))

    @property
    def params(self) -> dict:
        """The seeded synthetic weights (params.init_params), drawn on first use."""
        if self._params is None:
            from .params import init_params
            self._params = init_params(dict(self.cfg, dtype="fp32"), self.seed)
        return self._params

    # ---- inputs: synthetic, keyed by the GLOBAL image index (params.synthetic_batch) ----
    def inputs(self, first: int, count: int) -> dict:
        from .params import synthetic_batch
        cfg = dict(self.cfg, image_size=self.run["image"]) if self.tiled else self.cfg
        return synthetic_batch(cfg, self.seed, first, count)

    def cond(self, first: int, count: int):
        import torch
        c = torch.from_numpy(self.inputs(first, count)["cond"])
        return c if self.device is None else c.to(self.device)

    # ---- decode ----
    def decode_call(self, first: int, count: int):
        """One sampler call: images [first, first + count) -> [count, C, H, W]."""
        cond = self.cond(first, count)
        if self.tiled:
            return self.sampler.sample_tiled(cond, self.steps, overlap=self.run["overlap"], seed=self.seed, first_image=first)
        return self.sampler.sample(cond, self.steps, seed=self.seed, first_image=first)

    def decode(self) -> dict:
        """{global index: image} for this rank's shard, in calls of images_per_call."""
        return decode_shard(self.decode_call, self.total, self.rank, self.world, self.images_per_call)


def decode(total_images: int | None, cfg_name: str, **kw) -> dict:
    """Decode this rank's shard of `total_images` images of BASELINE.json configuration `cfg_name` (see ShardJob)."""
    return ShardJob(total_images, cfg_name, **kw).decode()
