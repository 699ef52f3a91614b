"""The build-defined contract for the decode hot path: UNet config dict, schedule
config dict and the five named configurations of BASELINE.json.

Reference: the snapshot at /root/reference holds no source (README.md is 0 bytes,
.gitignore:1-27 is the only content), so the "same UNet config dict, same
sample(cond, steps) entry point" that BASELINE.json's north_star names has no
referent there; SURVEY.md Appendix A fixes it and this file is that contract.
"""
from __future__ import annotations

import copy

UNET_DEFAULTS = dict(
    image_size=256,
    in_channels=3,
    out_channels=3,
    base_channels=128,
    channel_mult=(1, 1, 2, 2, 4, 4),
    num_res_blocks=2,
    groups=32,
    head_dim=64,
    attn_resolutions=(16,),
    cond_mode="concat",          # "concat" | "cross_attn"
    cond_channels=3,             # concat mode: channels of `cond` (nearest-resized, concatenated to x_t)
    context_dim=320,             # cross_attn mode: token width of `cond` [B, L, D]
    cross_attn_resolutions=(),   # cross_attn mode: feature resolutions that get a cross-attention block
    dtype="fp32",
)

SCHEDULE_DEFAULTS = dict(T=1000, kind="linear", beta_start=1e-4, beta_end=2e-2)


def unet_config(**overrides) -> dict:
    """Return a full UNet config dict (defaults + overrides), validated."""
    cfg = copy.deepcopy(UNET_DEFAULTS)
    unknown = set(overrides) - set(cfg)
    if unknown:
        raise KeyError(f"unknown UNet config keys: {sorted(unknown)}")
    cfg.update(overrides)
    return validate_unet_config(cfg)


def validate_unet_config(cfg: dict) -> dict:
    cfg = dict(UNET_DEFAULTS, **cfg)
    cfg["channel_mult"] = tuple(int(m) for m in cfg["channel_mult"])
    cfg["attn_resolutions"] = tuple(int(r) for r in cfg["attn_resolutions"])
    cfg["cross_attn_resolutions"] = tuple(int(r) for r in cfg["cross_attn_resolutions"])
    ch, g = cfg["base_channels"], cfg["groups"]
    if cfg["cond_mode"] not in ("concat", "cross_attn"):
        raise ValueError(f"cond_mode must be 'concat' or 'cross_attn', got {cfg['cond_mode']!r}")
    if cfg["dtype"] not in ("fp32", "fp16", "bf16"):
        raise ValueError(f"dtype must be 'fp32', 'fp16' or 'bf16', got {cfg['dtype']!r}")
    nlev = len(cfg["channel_mult"])
    if cfg["image_size"] % (1 << (nlev - 1)):
        raise ValueError("image_size must be divisible by 2**(levels-1)")
    res = cfg["image_size"]
    for lvl, m in enumerate(cfg["channel_mult"]):
        c = ch * m
        if c % g:
            raise ValueError(f"channels {c} not divisible by groups {g}")
        has_attn = (res in cfg["attn_resolutions"] or res in cfg["cross_attn_resolutions"] or lvl == nlev - 1)
        if has_attn and c % cfg["head_dim"]:
            raise ValueError(f"channels {c} (resolution {res}) not divisible by head_dim {cfg['head_dim']}")
        res //= 2
    if cfg["cond_mode"] == "cross_attn":
        cfg["cond_channels"] = 0
    return cfg


def named_config(name: str) -> tuple[dict, dict]:
    """(unet_cfg, run_cfg) for BASELINE.json configs[0..4] ("cfg1".."cfg5").

    run_cfg: batch, steps, method (cfg5 also: image = full image side, decoded as 256^2 tiles).
    """
    if name == "cfg1":   # 32x32x3 CIFAR-shaped, 64-ch UNet, 50 DDIM steps, batch 1
        return (unet_config(image_size=32, base_channels=64, channel_mult=(1, 2, 2, 2),
                            attn_resolutions=(16,)),
                dict(batch=1, steps=50, method="ddim"))
    if name in ("cfg2", "cfg3"):   # 256x256x3, 128-ch UNet, self-attn at 16^2, 100 DDIM steps
        return (unet_config(),
                dict(batch=16 if name == "cfg2" else 128, steps=100, method="ddim"))
    if name == "cfg4":   # 512x512x3, 192-ch UNet, cross-attn on latent tokens, 250 DDPM steps
        return (unet_config(image_size=512, base_channels=192, cond_mode="cross_attn",
                            attn_resolutions=(16,), cross_attn_resolutions=(32, 16),
                            context_dim=320),
                dict(batch=8, steps=250, method="ddpm"))
    if name == "cfg5":   # 1024x1024x3 tiled decode through the 256^2 UNet, fp16 storage / fp16 MFMA, 50 DDIM steps
        return (unet_config(dtype="fp16"),
                dict(batch=64, steps=50, method="ddim", image=1024, overlap=64))
    raise KeyError(name)
