"""Oracle UNet forward (U0-U9 of SURVEY.md section 8a).  TEST INFRASTRUCTURE ONLY.

Stock torch.nn.functional on the CPU, NCHW, no custom kernels, written as one plain
function that walks the architecture from the config dict independently of the product's
graph.py.  PARITY UNPINNED: the reference snapshot holds no model code (README.md 0 bytes);
the architecture is the one SURVEY.md Appendix A fixes:

  ResBlock(ci->co): h = conv3x3(silu(gn(x))); h += linear(silu(temb))[:, :, None, None];
                    h = conv3x3(silu(gn(h))); return h + (conv1x1(x) if ci != co else x)
  Attn(c):          x + proj(softmax(q k^T / sqrt(d)) v), q,k,v = qkv(gn(x)), heads = c // head_dim
  CrossAttn(c):     x + proj(softmax(q k^T / sqrt(d)) v), q = q(gn(x)), k,v = kv(context)
  Downsample: conv3x3 stride 2 pad 1.  Upsample: nearest x2 then conv3x3.
  temb: [sin(t f), cos(t f)], f_k = exp(-ln(1e4) k / (half-1)); Linear, SiLU, Linear.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def timestep_embedding_ref(t: torch.Tensor, dim: int, dtype=torch.float32) -> torch.Tensor:
    half = dim // 2
    k = torch.arange(half, dtype=torch.float64)
    freqs = torch.exp(-math.log(10000.0) * k / (half - 1))
    arg = t.to(torch.float64)[:, None] * freqs[None, :]
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=1).to(dtype)


def _p(params, name, dtype):
    v = params[name]
    if not isinstance(v, torch.Tensor):
        v = torch.from_numpy(v)
    return v.to(dtype)


def unet_forward_ref(cfg: dict, params: dict, x: torch.Tensor, t: torch.Tensor, cond: torch.Tensor,
                     dtype=torch.float32, taps: dict | None = None) -> torch.Tensor:
    """eps = UNet(x_t, t, cond).  x [B,3,H,W]; t [B] int; cond [B,Cc,hc,wc] (concat) or [B,L,D].

    `taps`, if given, is filled with named intermediate tensors: per block ("conv_in", "down.i.j", "mid", "up.i.j") and per
    convolution layer under the layer's parameter name, taken after everything the HIP launch of that name fuses (+ temb for
    conv1, + skip for conv2, + x for attention projections).
    """
    ch = cfg["base_channels"]
    groups = cfg["groups"]
    nrb = cfg["num_res_blocks"]
    mults = tuple(cfg["channel_mult"])
    hd = cfg["head_dim"]
    cross = cfg["cond_mode"] == "cross_attn"
    W = lambda n: _p(params, n, dtype)   # noqa: E731
    x = x.to(dtype)

    def gn(h, name):
        return F.group_norm(h, groups, W(name + ".weight"), W(name + ".bias"), eps=1e-5)

    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v

    def conv(h, name, stride=1, pad=1):
        return tap(name, F.conv2d(h, W(name + ".weight"), W(name + ".bias"), stride=stride, padding=pad))

    temb = timestep_embedding_ref(t, ch, dtype)
    temb = F.linear(temb, W("temb.0.weight"), W("temb.0.bias"))
    temb = F.linear(F.silu(temb), W("temb.2.weight"), W("temb.2.bias"))
    temb_act = F.silu(temb)

    def res(h, name):
        ci = h.shape[1]
        co = params[name + ".conv1.weight"].shape[0]
        y = conv(F.silu(gn(h, name + ".norm1")), name + ".conv1")
        y = tap(name + ".conv1", y + F.linear(temb_act, W(name + ".temb.weight"), W(name + ".temb.bias"))[:, :, None, None])
        y = conv(F.silu(gn(y, name + ".norm2")), name + ".conv2")
        sk = conv(h, name + ".skip", pad=0) if ci != co else h
        return tap(name + ".conv2", y + sk)      # (taps name a tensor after everything the HIP layer of that name fuses: + temb, + skip)

    def mha(q, k, v):
        # q [B, C, Nq], k/v [B, C, Nk] -> [B, C, Nq]; heads = C // hd
        B, C, Nq = q.shape
        nh = C // hd
        q = q.reshape(B, nh, hd, Nq)
        k = k.reshape(B, nh, hd, -1)
        v = v.reshape(B, nh, hd, -1)
        s = torch.einsum("bhdq,bhdk->bhqk", q, k) * (1.0 / math.sqrt(hd))
        p = torch.softmax(s, dim=-1)
        o = torch.einsum("bhqk,bhdk->bhdq", p, v)
        return o.reshape(B, C, Nq)

    def attn(h, name):
        B, C, Hh, Ww = h.shape
        qkv = conv(gn(h, name + ".norm"), name + ".qkv", pad=0).reshape(B, 3 * C, Hh * Ww)
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        o = mha(q, k, v).reshape(B, C, Hh, Ww)
        return tap(name + ".proj", h + conv(o, name + ".proj", pad=0))

    def xattn(h, name):
        B, C, Hh, Ww = h.shape
        q = conv(gn(h, name + ".norm"), name + ".q", pad=0).reshape(B, C, Hh * Ww)
        kv = F.linear(cond.to(dtype), W(name + ".kv.weight"), W(name + ".kv.bias"))   # [B, L, 2C]
        kv = kv.transpose(1, 2)
        o = mha(q, kv[:, :C], kv[:, C:]).reshape(B, C, Hh, Ww)
        return tap(name + ".proj", h + conv(o, name + ".proj", pad=0))

    def extras(h, prefix, r):
        if r in cfg["attn_resolutions"]:
            h = attn(h, prefix + ".attn")
        if cross and r in cfg["cross_attn_resolutions"]:
            h = xattn(h, prefix + ".xattn")
        return h

    if cross:
        h = x
    else:
        c = F.interpolate(cond.to(dtype), size=x.shape[-2:], mode="nearest")
        h = torch.cat([x, c], dim=1)
    h = conv(h, "conv_in")
    if taps is not None:
        taps["conv_in"] = h
    skips = [h]
    r = cfg["image_size"]
    for i in range(len(mults)):
        for j in range(nrb):
            h = res(h, f"down.{i}.{j}.res")
            h = extras(h, f"down.{i}.{j}", r)
            skips.append(h)
            if taps is not None:
                taps[f"down.{i}.{j}"] = h
        if i != len(mults) - 1:
            h = conv(h, f"down.{i}.ds", stride=2)
            skips.append(h)
            r //= 2
    h = res(h, "mid.0.res")
    h = attn(h, "mid.1.attn")
    if cross:
        h = xattn(h, "mid.1.xattn")
    h = res(h, "mid.2.res")
    if taps is not None:
        taps["mid"] = h
    for i in reversed(range(len(mults))):
        for j in range(nrb + 1):
            h = res(torch.cat([h, skips.pop()], dim=1), f"up.{i}.{j}.res")
            h = extras(h, f"up.{i}.{j}", r)
            if taps is not None:
                taps[f"up.{i}.{j}"] = h
        if i != 0:
            h = conv(F.interpolate(h, scale_factor=2, mode="nearest"), f"up.{i}.us")
            r *= 2
    assert not skips
    return conv(F.silu(gn(h, "out.norm")), "out.conv")
