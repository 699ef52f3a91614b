"""Sampler(unet, schedule, method).sample(cond, steps) -- the reverse-diffusion loop (S2 of SURVEY.md 8a).

The entry point BASELINE.json north_star names.  Per call: x_T is drawn on the device from the counter
generator (stream keyed by the GLOBAL image index, so results do not depend on how a batch is sharded
over GPUs), then `steps` iterations of {UNet forward (recorded launch list), one fused update kernel}.
The loop body issues only C-ABI launches on the current stream: no allocation, no host sync, no
device-to-host copy until the result is exported.

The reference snapshot has no sampler to cite (README.md: 0 bytes); equations: schedule.py.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import _abi, ops
from .rng import STREAM_STEP0, STREAM_XT
from .schedule import make_schedule, step_coefficients
from .unet import UNet, load_cond


@dataclass
class _Run:
    """One decode in flight (Sampler.begin): the plan whose buffers hold the state, the per-step coefficients, the
    update kernel's argument struct, and the optional hipGraph of the forward."""
    plan: object
    coefs: list
    upd: object
    graph: object
    steps: int
    channels: int
    t_host: int


class Sampler:
    def __init__(self, unet: UNet, schedule: dict | None = None, method: str = "ddim", *,
                 eta: float = 0.0, clip_x0: bool = True, use_graph: bool = False):
        if method not in ("ddim", "ddpm"):
            raise ValueError(f"method must be 'ddim' or 'ddpm', got {method!r}")
        if eta != 0.0:
            raise ValueError("only eta = 0 (deterministic DDIM) is defined")
        self.unet, self.method, self.clip_x0 = unet, method, clip_x0
        self.use_graph = use_graph       # replay the UNet forward as one hipGraph launch (host-bound workloads)
        self._graphs = {}                # batch -> hipGraph of the forward; owned by THIS sampler (the plan stays eager)
        self.schedule = make_schedule(schedule)

    @torch.no_grad()
    def sample(self, cond: torch.Tensor, steps: int, *, seed: int = 0, first_image: int = 0,
               trace: list | None = None, x_T: torch.Tensor | None = None, on_finish=None) -> torch.Tensor:
        """cond [B,Cc,hc,wc] or [B,L,D] -> x_0 [B,C,H,W] in [-1,1] on the UNet's device.

        `first_image` is the global index of cond[0] (noise streams are keyed by global index).
        `x_T` [B,C,H,W], if given, replaces the generator's x_T (tiled decode: crops of one noise field).
        `on_finish(run)`, if given, is called after the last step with the decode's state still in the plan's buffers (e.g. the
        8-bit export straight from run.plan.xin: bitstream.decode_bitstreams) -- so no caller re-implements this loop.
        = begin() + `steps` x step() + finish(); bench.py times exactly these step() calls.
        """
        run = self.begin(cond, steps, seed=seed, first_image=first_image, x_T=x_T)
        for k in range(steps):
            self.step(run, k)
            if trace is not None:
                trace.append(run.plan.xin[..., :run.channels].permute(0, 3, 1, 2).clone())
        if on_finish is not None:
            on_finish(run)
        return self.finish(run)

    @torch.no_grad()
    def begin(self, cond: torch.Tensor, steps: int, *, seed: int = 0, first_image: int = 0,
              x_T: torch.Tensor | None = None) -> "_Run":
        """Load cond, draw x_T into the plan's state buffer and return the handle step() / finish() take.  One decode
        per (UNet, batch size) can be in flight at a time: the state lives in the plan's buffers."""
        net, cfg = self.unet, self.unet.cfg
        with torch.cuda.device(net.device):      # every launch targets the net's device, whatever the caller's is
            B, C = cond.shape[0], cfg["in_channels"]
            p = net.plan(B)
            graph = None
            if self.use_graph:
                if B not in self._graphs:
                    self._graphs[B] = p.capture()
                graph = self._graphs[B]
            load_cond(p, cfg, cond)
            if x_T is None:
                ops.gauss_fill(p.xin, C, seed, first_image, STREAM_XT)
            else:
                p.xin[..., :C] = x_T.to(net.device, torch.float32).permute(0, 2, 3, 1)
            upd = _abi.DiffusionUpdateArgs()
            upd.x, upd.x_ld, upd.eps, upd.eps_ld = p.xin.data_ptr(), p.xin.shape[-1], p.eps.data_ptr(), p.eps.shape[-1]
            upd.batch, upd.hw, upd.channels = B, p.xin.shape[1] * p.xin.shape[2], C
            upd.clip_x0, upd.seed, upd.first_image = int(self.clip_x0), seed, first_image
            return _Run(plan=p, coefs=step_coefficients(self.schedule, steps, self.method), upd=upd, graph=graph,
                        steps=steps, channels=C, t_host=-1)

    @torch.no_grad()
    def step(self, run: "_Run", k: int) -> None:
        """Reverse step k (0-based, execution order: k = 0 is t = tau_{S-1}): one UNet forward (recorded launch list or
        its hipGraph) + the fused update kernel.  Only C-ABI launches on the current stream of the net's device."""
        dev = self.unet.device
        with torch.cuda.device(dev):
            p, c, upd = run.plan, run.coefs[k], run.upd
            st = torch.cuda.current_stream(dev).cuda_stream
            if c.t != run.t_host:
                p.t.fill_(c.t)
                run.t_host = c.t
            if run.graph is not None:
                run.graph.replay()               # on the current stream (= st)
            else:
                p.run(st)
            upd.ca, upd.cb, upd.cx, upd.c0, upd.ce, upd.sigma = c.ca, c.cb, c.cx, c.c0, c.ce, c.sigma
            upd.noise_stream = STREAM_STEP0 + k
            _abi.call("diffusion_update_f32", upd, None, 0, st)

    @torch.no_grad()
    def finish(self, run: "_Run") -> torch.Tensor:
        with torch.cuda.device(self.unet.device):
            return ops.export_image(run.plan.xin, run.channels, -1.0, 1.0)

    def sample_tiled(self, cond: torch.Tensor, steps: int, **kw) -> torch.Tensor:
        """Decode an image larger than the UNet's native size tile by tile (tiling.py, S5)."""
        from .tiling import sample_tiled
        return sample_tiled(self, cond, steps, **kw)


def sample(unet: UNet, cond: torch.Tensor, steps: int, *, method: str = "ddim", schedule: dict | None = None,
           seed: int = 0, **kw) -> torch.Tensor:
    """Functional form of Sampler(unet, schedule, method).sample(cond, steps, seed=seed)."""
    return Sampler(unet, schedule, method).sample(cond, steps, seed=seed, **kw)
