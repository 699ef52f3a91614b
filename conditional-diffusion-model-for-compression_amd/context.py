"""Context front-end: the step immediately before `sample(cond, steps)` in a
conditional-diffusion codec -- the synthesis-side network that turns the (entropy-decoded) quantised latent
z [B, Cz, H/16, W/16] into the conditioning the UNet consumes:

    concat mode      cond [B, Cc, (H/16) 2^u, (W/16) 2^u]   (the UNet nearest-resizes it to H x W and concatenates it to x_t)
    cross_attn mode  tokens [B, L = (H/16)(W/16), D]          (the conv output in NHWC IS the token matrix: no copy)

Build-defined (the reference snapshot holds no model code: README.md is 0 bytes); small canonical stack:

    h = conv3x3(z)                                   ctx.in     Cz -> hidden
    h = h + conv3x3(silu(gn(h)))      x num_blocks   ctx.block{i}
    h = conv3x3(nearest2x(h))         x upsample     ctx.up{u}  (concat mode only)
    out = conv3x3(silu(gn(h)))                       ctx.out    hidden -> Cc | D

Every layer is a cdx_conv_f32 launch (GroupNorm + SiLU fused on load, residual and GroupNorm sums in the epilogue,
nearest-2x fused into the gather): the same kernels, tile choice and parity bar as the UNet.  The launch list is
recorded once per batch size and replayed; it runs ONCE per decode (cond does not depend on the timestep), so it adds
no host work to the sampling loop.  No CPU fallback.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np
import torch

from . import _abi, ops, rng
from .config import validate_unet_config

CONTEXT_DEFAULTS = dict(latent_channels=16, hidden=64, num_blocks=2, upsample=2, groups=32)
PARAM_B = (1 << 41)      # weight streams of the context net: a = PARAM_B + parameter index (UNet: rng.PARAM_A)


def context_config(**overrides) -> dict:
    cfg = dict(CONTEXT_DEFAULTS)
    unknown = set(overrides) - set(cfg)
    if unknown:
        raise KeyError(f"unknown context config keys: {sorted(unknown)}")
    cfg.update(overrides)
    if cfg["hidden"] % cfg["groups"] or cfg["latent_channels"] % 8 or cfg["hidden"] % 8:
        raise ValueError("hidden must be divisible by groups; latent_channels and hidden by 8")
    return cfg


def context_param_shapes(unet_cfg: dict, ctx_cfg: dict) -> dict:
    """name -> shape (torch layouts), in definition order."""
    ucfg = validate_unet_config(unet_cfg)
    cz, hd = ctx_cfg["latent_channels"], ctx_cfg["hidden"]
    cross = ucfg["cond_mode"] == "cross_attn"
    cout = ucfg["context_dim"] if cross else ucfg["cond_channels"]
    P = {}

    def conv(name, ci, co):
        P[f"{name}.weight"] = (co, ci, 3, 3)
        P[f"{name}.bias"] = (co,)

    def norm(name, c):
        P[f"{name}.weight"] = (c,)
        P[f"{name}.bias"] = (c,)

    conv("ctx.in", cz, hd)
    for i in range(ctx_cfg["num_blocks"]):
        norm(f"ctx.block{i}.norm", hd)
        conv(f"ctx.block{i}.conv", hd, hd)
    for u in range(0 if cross else ctx_cfg["upsample"]):
        conv(f"ctx.up{u}", hd, hd)
    norm("ctx.out.norm", hd)
    conv("ctx.out", hd, cout)
    return P


def init_context_params(unet_cfg: dict, ctx_cfg: dict, seed: int = 0, affine_jitter: float = 0.0) -> dict:
    """Seeded synthetic weights (fan-in-scaled uniform, GroupNorm gamma 1 / beta 0), as params.init_params."""
    params = {}
    for pidx, (name, shape) in enumerate(context_param_shapes(unet_cfg, ctx_cfg).items()):
        n = int(np.prod(shape))
        key = rng.stream_key(seed, PARAM_B + pidx, 0)
        if ".norm" in name:
            base = 1.0 if name.endswith(".weight") else 0.0
            v = base + affine_jitter * (2.0 * rng.uniform(key, n) - 1.0) if affine_jitter else np.full(n, base)
        else:
            wshape = shape if name.endswith(".weight") else context_param_shapes(unet_cfg, ctx_cfg)[name[:-5] + ".weight"]
            bound = 1.0 / math.sqrt(int(np.prod(wshape[1:])))
            v = (2.0 * rng.uniform(key, n) - 1.0) * bound
        params[name] = np.asarray(v, np.float32).reshape(shape)
    return params


def synthetic_latent(ctx_cfg: dict, image_size: int, seed: int, first_image: int, count: int) -> np.ndarray:
    """z [count, Cz, H/16, W/16] ~ N(0, 1), keyed by the GLOBAL image index (stream rng.STREAM_COND)."""
    cz, h = ctx_cfg["latent_channels"], image_size // 16
    return np.stack([rng.normal(rng.stream_key(seed, first_image + k, rng.STREAM_COND), cz * h * h).reshape(cz, h, h)
                     for k in range(count)])


class _CtxPlan:
    def __init__(self, net: "ContextNet", batch: int, hz: int, wz: int):
        dev, cc, g = net.device, net.ctx_cfg, net.ctx_cfg["groups"]
        self.calls, self._keep = [], []
        new = lambda *s: self._hold(torch.empty(*s, device=dev, dtype=torch.float32))      # noqa: E731
        self.z = self._hold(torch.zeros(batch, hz, wz, cc["latent_channels"], device=dev))
        stats_of = {}
        # range bookkeeping of the split tiles, as in unet._Plan: one int32 word per (tensor, image), zeroed per run
        split = net.split
        self.amax_arena = self._hold(torch.zeros(8 + net.ups + cc["num_blocks"], batch, _abi.AMAX_WORDS, dtype=torch.int32, device=dev))
        amax_slot, need_amax, produced = {}, set(), {}

        def amax_of(t):
            if t.data_ptr() not in amax_slot:
                amax_slot[t.data_ptr()] = self.amax_arena[len(amax_slot)]
            return amax_slot[t.data_ptr()]

        if split:
            self.calls.append((_abi.lib().cdx_fill_u32, _abi.FillU32Args(self.amax_arena.data_ptr(), self.amax_arena.numel(), 0)))

        def conv(name, src, *, norm=None, upsample=False, residual=None, normed_later=False, out_ld=None):
            pc = net.convs[name]
            B, h, w, _ = src.shape
            ho, wo = (2 * h, 2 * w) if upsample else (h, w)
            out = new(B, ho, wo, out_ld or pc.cout)
            if out.shape[-1] != pc.cout:
                out.zero_()
            gnp = (new(batch, src.shape[-1]), new(batch, src.shape[-1])) if norm is not None else None
            a = ops.conv_args(pc, src, None, out, upsample=upsample, gn=gnp, silu=norm is not None, residual=residual, out_ld=out_ld)
            if split and norm is not None:
                a.gn_exp = _abi.gn_act_exp(net.host[norm + ".weight"], net.host[norm + ".bias"], g, h * w)
                a.flags |= _abi.CONV_GN_EXP
                if _abi.lib().cdx_conv_select_tile(ctypes.byref(a)) != _abi.TILE_SPLIT:
                    a.gn_exp, a.flags = 0, a.flags & ~_abi.CONV_GN_EXP
            elif split:
                a.src_amax0 = amax_of(src).data_ptr()
                if _abi.lib().cdx_conv_select_tile(ctypes.byref(a)) == _abi.TILE_SPLIT:
                    if src.data_ptr() in produced:
                        need_amax.add(src.data_ptr())
                    else:      # the latent: caller data
                        self.calls.append((_abi.lib().cdx_amax_f32, _abi.AmaxArgs(src.data_ptr(), src.shape[-1], B, h * w, src.shape[-1],
                                                                                 amax_of(src).data_ptr())))
                else:
                    a.src_amax0 = None
            if norm is not None:
                ga = ops.gn_finalize_args(stats_of[src.data_ptr()], None, h * w, net.dev[norm + ".weight"], net.dev[norm + ".bias"],
                                          g, gnp[0], gnp[1], out_exp=a.gn_exp)
                self.calls.append((_abi.lib().cdx_gn_finalize_f32, ga))
            if normed_later:      # (asked AFTER the range fields are set: they decide the tile, the tile the slots)
                stats_of[out.data_ptr()] = self._hold(ops.conv_stats_buffer(a, dev))
            produced[out.data_ptr()] = a
            self.calls.append((_abi.lib().cdx_conv_f32, a))
            return out

        h = conv("ctx.in", self.z, normed_later=True)
        for i in range(cc["num_blocks"]):
            h = conv(f"ctx.block{i}.conv", h, norm=f"ctx.block{i}.norm", residual=h, normed_later=True)
        for u in range(net.ups):
            h = conv(f"ctx.up{u}", h, upsample=True, normed_later=True)
        pc = net.convs["ctx.out"]
        self.out = conv("ctx.out", h, norm="ctx.out.norm", out_ld=(pc.cout + 3) // 4 * 4)
        self.cout = pc.cout
        for k in need_amax:
            produced[k].amax_out = amax_slot[k].data_ptr()

    def _hold(self, t):
        self._keep.append(t)
        return t

    def run(self):
        st = torch.cuda.current_stream(self.z.device).cuda_stream
        for fn, a in self.calls:
            rc = fn(ctypes.byref(a), None, 0, st)
            if rc:
                _abi.check(rc, fn.__name__)


class ContextNet:
    """ctx = ContextNet(unet_cfg[, ctx_cfg, params]);  cond = ctx(z)  ->  what Sampler.sample(cond, steps) takes."""

    def __init__(self, unet_cfg: dict, ctx_cfg: dict | None = None, params: dict | None = None, *, seed: int = 0,
                 device="cuda", split: bool = True):
        _abi.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("ContextNet (HIP backend) needs a GPU; there is no CPU fallback in the product path")
        self.unet_cfg = validate_unet_config(unet_cfg)
        self.ctx_cfg = context_config(**(ctx_cfg or {}))
        self.cross = self.unet_cfg["cond_mode"] == "cross_attn"
        self.ups = 0 if self.cross else self.ctx_cfg["upsample"]
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        shapes = context_param_shapes(self.unet_cfg, self.ctx_cfg)
        if params is None:
            params = init_context_params(self.unet_cfg, self.ctx_cfg, seed)
        P = {k: np.asarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, np.float32) for k, v in params.items()}
        for k, shp in shapes.items():
            if k not in P or tuple(P[k].shape) != tuple(shp):
                raise ValueError(f"context param {k}: expected shape {shp}")
        self.convs, self.dev, self.host = {}, {}, {}
        for name, shp in shapes.items():
            if name.endswith(".weight") and len(shp) == 4:
                base = name[:-7]
                self.convs[base] = ops.PackedConv(P[name], P[base + ".bias"], shp[1], 0, self.device, split=split, up=base.startswith("ctx.up"))
            elif ".norm" in name:
                self.dev[name] = torch.from_numpy(np.ascontiguousarray(P[name])).to(self.device)
                self.host[name] = P[name]
        self.split = split
        self._plans = {}

    @torch.no_grad()
    def forward(self, z: torch.Tensor) -> torch.Tensor:
        """z [B, Cz, hz, wz] -> cond [B, Cc, hz 2^u, wz 2^u] (concat mode) or tokens [B, hz wz, D] (cross_attn mode)."""
        B, cz, hz, wz = z.shape
        assert cz == self.ctx_cfg["latent_channels"], z.shape
        with torch.cuda.device(self.device):
            key = (B, hz, wz)
            if key not in self._plans:
                self._plans[key] = _CtxPlan(self, B, hz, wz)
            p = self._plans[key]
            p.z.copy_(z.to(self.device, torch.float32).permute(0, 2, 3, 1))
            p.run()
            out = p.out[..., :p.cout]
            if self.cross:
                return out.reshape(B, hz * wz, p.cout).contiguous()
            return out.permute(0, 3, 1, 2).contiguous()

    __call__ = forward


@torch.no_grad()
def decode_latent(sampler, ctx: ContextNet, z: torch.Tensor, steps: int, **kw) -> torch.Tensor:
    """latent -> conditioning (once) -> reverse diffusion: x_0 = sampler.sample(ctx(z), steps)."""
    return sampler.sample(ctx(z), steps, **kw)
