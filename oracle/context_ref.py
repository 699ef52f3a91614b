"""Oracle restatement of the context front-end (latent -> conditioning).  TEST INFRASTRUCTURE ONLY.

Stock torch.nn.functional on the CPU, written independently of the product's context.py (which records cdx_conv_f32
launches).  PARITY UNPINNED by the reference (its snapshot holds no model code: README.md is 0 bytes); the definition:

    h = conv3x3(z);  h = h + conv3x3(silu(gn(h)))  x num_blocks;  h = conv3x3(nearest2x(h))  x upsample (concat mode);
    out = conv3x3(silu(gn(h)));   concat mode: cond = out [B, Cc, h, w];  cross_attn mode: tokens = out as [B, h w, D]
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def context_forward_ref(unet_cfg: dict, ctx_cfg: dict, params: dict, z: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    W = lambda n: torch.as_tensor(params[n]).to(dtype)      # noqa: E731
    g = ctx_cfg["groups"]
    cross = unet_cfg["cond_mode"] == "cross_attn"

    def conv(h, name):
        return F.conv2d(h, W(name + ".weight"), W(name + ".bias"), padding=1)

    def act(h, name):
        return F.silu(F.group_norm(h, g, W(name + ".weight"), W(name + ".bias"), eps=1e-5))

    h = conv(z.to(dtype), "ctx.in")
    for i in range(ctx_cfg["num_blocks"]):
        h = h + conv(act(h, f"ctx.block{i}.norm"), f"ctx.block{i}.conv")
    for u in range(0 if cross else ctx_cfg["upsample"]):
        h = conv(F.interpolate(h, scale_factor=2, mode="nearest"), f"ctx.up{u}")
    out = conv(act(h, "ctx.out.norm"), "ctx.out")
    if cross:
        B, D, hh, ww = out.shape
        return out.permute(0, 2, 3, 1).reshape(B, hh * ww, D)
    return out
