#!/usr/bin/env python3
"""Randomised UNet-forward parity sweep: random architectures (sizes, channel multipliers, attention placement, conditioning
mode, fp32 / fp16 storage) through the HIP path against the float64 oracle on the CPU.  Test infrastructure (lives under
tests/ because it imports oracle/).  usage: python tests/fuzz_unet.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cdx
import oracle

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = done = 0
while done < ncases:
    size = int(rng.choice([16, 32, 64]))
    base = int(rng.choice([32, 64, 96]))
    nlev = int(rng.integers(2, 5))
    mult = tuple(int(rng.choice([1, 2])) * (2 if i >= 2 and rng.integers(0, 2) else 1) for i in range(nlev))
    if size >> (nlev - 1) < 4:
        continue
    res = [size >> i for i in range(nlev)]
    attn = tuple(r for r in res if r <= 32 and rng.integers(0, 2))
    cross = bool(rng.integers(0, 3) == 0)
    over = dict(image_size=size, base_channels=base, channel_mult=mult, attn_resolutions=attn, num_res_blocks=int(rng.integers(1, 3)))
    if cross:
        over.update(cond_mode="cross_attn", cross_attn_resolutions=tuple(r for r in res if rng.integers(0, 2)) or (res[-1],), context_dim=int(rng.choice([64, 96])))
    half = bool(rng.integers(0, 3) == 0)
    if half:
        over["dtype"] = "fp16"
    try:
        cfg = cdx.unet_config(**over)
    except (ValueError, KeyError):
        continue                        # e.g. a level whose channels are not a multiple of head_dim
    done += 1
    params = cdx.init_params(cfg, seed=100 + done, affine_jitter=0.1, out_gain=1.0)
    B = 2
    cond = torch.from_numpy(cdx.synthetic_batch(cfg, 7, 0, B)["cond"])
    x = torch.randn(B, 3, size, size, generator=torch.Generator().manual_seed(done))
    t = torch.tensor([int(rng.integers(0, 1000)), int(rng.integers(0, 1000))])
    want = oracle.unet_forward_ref(dict(cfg, dtype="fp32"), params, x, t, cond, dtype=torch.float64)
    got = cdx.UNet(cfg, params).forward(x.cuda(), t.cuda(), cond.cuda()).float().cpu().double()
    err = (got - want).abs().max().item() if not torch.isnan(got).any() else float("inf")
    tol = (1e-2 if half else 2e-5) * max(1.0, want.abs().max().item())
    ok = err <= tol
    bad += not ok
    print("case", done, "ok " if ok else "BAD", "err %.3e" % err, over)
print("fuzz_unet:", bad, "bad of", ncases)
