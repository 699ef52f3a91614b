"""BASELINE.json configs[3] and configs[4] under test on the GPU ("configs_untested").

configs[3]: 512x512x3, 192-ch UNet with cross-attention on compressed-latent tokens (L = 1024, D = 320), 250 DDPM
            steps, batch 8 -- the REAL network (named_config("cfg4")): one image, single forward against the float64
            oracle, a few ancestral steps against the float32 oracle through both PSNR gates, plus the same UNet at
            image_size 128 (same widths: every 192/384/768-channel tile path) at batch 2.
configs[4]: 1024x1024x3 decoded as 25 overlapping 256^2 tiles (overlap 64) through the fp16-storage UNet, 50 DDIM
            steps: finite, clamped, equal to the per-tile decodes blended by the oracle's float64 blend, exact in tile
            cores, and within the stated fp16 tolerance of the float32 HIP decode of the same image.

Tolerances: float32 as in test_e2e_gpu.py (forward <= 6e-6 x scale vs float64 (8e-6 vs the float32 oracle at full size); PSNR(hip, oracle) >= 80 dB;
|dPSNR vs target| <= 0.01 dB); fp16 as in test_fp16_gpu.py (PSNR >= 45 dB, |dPSNR| <= 0.01 dB).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def psnr(a, b):
    mse = ((a.double() - b.double()) ** 2).mean().item()
    return float("inf") if mse == 0 else 10.0 * math.log10(4.0 / mse)


@pytest.fixture(scope="module")
def cdx_mod(lib):
    import cdx
    assert torch.cuda.is_available()
    return cdx


def _xt(cdx_mod, cfg, seed, B):
    H = cfg["image_size"]
    return torch.from_numpy(np.stack([cdx_mod.rng.normal(cdx_mod.rng.stream_key(seed, i, 1), 3 * H * H).reshape(3, H, H)
                                      for i in range(B)]))


def test_cfg4_widths_reduced_image_forward_and_ddpm(cdx_mod, record):
    """cfg4's UNet (192/384/768 channels, self-attention at 16^2, cross-attention at 32^2 and 16^2, D = 320) at
    image_size 128 (L = 64 tokens), batch 2: forward vs float64, then 4 ancestral steps vs the float32 oracle."""
    import oracle
    cfg, run = cdx_mod.named_config("cfg4")
    cfg = cdx_mod.unet_config(**dict(cfg, image_size=128))
    assert cfg["base_channels"] == 192 and cfg["cond_mode"] == "cross_attn" and run["method"] == "ddpm"
    B = 2
    params = cdx_mod.init_params(cfg, seed=4, affine_jitter=0.1, out_gain=1.0)
    sb = cdx_mod.synthetic_batch(cfg, 4, 0, B)
    cond = torch.from_numpy(sb["cond"])
    assert tuple(cond.shape) == (B, 64, 320)
    x, t = _xt(cdx_mod, cfg, 4, B), torch.tensor([930, 41])
    want = oracle.unet_forward_ref(cfg, params, x, t, cond, dtype=torch.float64)
    net = cdx_mod.UNet(cfg, params)
    got = net.forward(x.cuda(), t.cuda(), cond.cuda()).cpu()
    err, scale = (got.double() - want).abs().max().item(), want.abs().max().item()
    record("cfg4_128_forward", hip_vs_fp64=err, scale=scale)
    assert err <= 6e-6 * max(1.0, scale), f"max err {err:.3e} (scale {scale:.3e})"      # (measured 2.6e-6: twice that)
    # sampler: the benchmark's weights (out_gain 0.05), DDPM noise every step
    params = cdx_mod.init_params(cfg, seed=4)
    tgt = torch.from_numpy(sb["target"])
    got = cdx_mod.Sampler(cdx_mod.UNet(cfg, params), method="ddpm").sample(cond.cuda(), 4, seed=4).cpu()
    want = oracle.sample_ref(cfg, params, cond, 4, seed=4, method="ddpm")
    record("cfg4_128_ddpm4", psnr_hip_vs_oracle=psnr(got, want), dpsnr=abs(psnr(got, tgt) - psnr(want, tgt)))
    assert psnr(got, want) >= 80.0 and abs(psnr(got, tgt) - psnr(want, tgt)) <= 0.01


def test_cfg4_full_size_vs_live_oracle(cdx_mod, record):
    """BASELINE.json configs[3] itself: 512x512, 192-ch, cross-attention on L = 1024 x D = 320 tokens, bench.py's weights
    (seed 0).  One image: (a) a single forward at out_gain = 1 against the float32 oracle run on this box's CPU (float64
    costs minutes at this size; the float32 oracle's own error vs float64 is ~1e-6 x scale, measured at image_size 128);
    (b) 2 DDPM steps against the float32 oracle, both PSNR gates."""
    import oracle
    cfg, run = cdx_mod.named_config("cfg4")
    assert cfg["image_size"] == 512 and cfg["context_dim"] == 320 and cfg["cross_attn_resolutions"] == (32, 16)
    sb = cdx_mod.synthetic_batch(cfg, 0, 0, 1)
    cond, tgt = torch.from_numpy(sb["cond"]), torch.from_numpy(sb["target"])
    assert tuple(cond.shape) == (1, 1024, 320)
    params = cdx_mod.init_params(cfg, seed=0, out_gain=1.0)
    x, t = _xt(cdx_mod, cfg, 0, 1), torch.tensor([777])
    want = oracle.unet_forward_ref(cfg, params, x, t, cond)
    got = cdx_mod.UNet(cfg, params).forward(x.cuda(), t.cuda(), cond.cuda()).cpu()
    err, scale = (got.double() - want.double()).abs().max().item(), want.abs().max().item()
    record("cfg4_full_forward", hip_vs_cpu_fp32=err, scale=scale)
    assert err <= 8e-6 * max(1.0, scale), f"max err {err:.3e} (scale {scale:.3e})"      # (measured 3.1-4.1e-6 against the FLOAT32 oracle, which is itself ~1e-6 off: twice that)
    del want, got
    torch.cuda.empty_cache()
    params = cdx_mod.init_params(cfg, seed=0)
    got = cdx_mod.Sampler(cdx_mod.UNet(cfg, params), method="ddpm").sample(cond.cuda(), 2, seed=0).cpu()
    want = oracle.sample_ref(cfg, params, cond, 2, seed=0, method="ddpm")
    record("cfg4_full_ddpm2", psnr_hip_vs_oracle=psnr(got, want), dpsnr=abs(psnr(got, tgt) - psnr(want, tgt)),
           max_err=(got - want).abs().max().item())
    assert psnr(got, want) >= 80.0 and abs(psnr(got, tgt) - psnr(want, tgt)) <= 0.01


def test_cfg4_full_size_properties_batch2(cdx_mod):
    """Size-independent properties at the full cfg4 shape, batch 2, 3 DDPM steps: finite, clamped, deterministic, and
    image 1 is bit-identical decoded alone as global image 1 (noise streams are keyed by the global index)."""
    cfg, _ = cdx_mod.named_config("cfg4")
    net = cdx_mod.UNet(cfg, seed=0)
    cond = torch.from_numpy(cdx_mod.synthetic_batch(cfg, 0, 0, 2)["cond"]).cuda()
    s = cdx_mod.Sampler(net, method="ddpm")
    x2 = s.sample(cond, 3, seed=0)
    x1 = s.sample(cond[1:2].contiguous(), 3, seed=0, first_image=1)
    assert torch.isfinite(x2).all() and x2.abs().max().item() <= 1.0
    assert torch.equal(x2[1:2], x1)
    assert torch.equal(x2, s.sample(cond, 3, seed=0))


def test_cfg5_full_1024_tiled_fp16_decode(cdx_mod, record):
    """BASELINE.json configs[4]: one 1024x1024 image = 25 tiles of 256^2 (overlap 64) through the fp16 UNet, 50 DDIM steps."""
    import oracle
    cfg16, run = cdx_mod.named_config("cfg5")
    assert run["image"] == 1024 and run["overlap"] == 64 and run["steps"] == 50 and cfg16["dtype"] == "fp16"
    T, S, ov = cfg16["image_size"], run["image"], run["overlap"]
    ys, xs = cdx_mod.tile_plan(S, S, T, ov)
    assert len(ys) == len(xs) == 5 and ys == oracle.origins_ref(S, T, ov)
    params = cdx_mod.init_params(dict(cfg16, dtype="fp32"), seed=0)
    # conditioning of the whole image: [1, 3, 64, 64] (one cell per 16x16 pixels), from the 1024^2 synthetic target
    big = cdx_mod.synthetic_batch(dict(cfg16, image_size=S, dtype="fp32"), 0, 0, 1)
    cond = torch.from_numpy(big["cond"]).cuda()
    assert tuple(cond.shape) == (1, 3, S // 16, S // 16)
    sampler = cdx_mod.Sampler(cdx_mod.UNet(cfg16, params))
    out = sampler.sample_tiled(cond, run["steps"], overlap=ov, seed=0, tiles_per_call=16)
    assert tuple(out.shape) == (1, 3, S, S)
    assert torch.isfinite(out).all() and out.abs().max().item() <= 1.0
    # the same 25 tiles decoded one call each (batch 1: image bits do not depend on the batch), blended by the ORACLE's blend
    full = torch.zeros(1, S, S, 4, device="cuda")
    cdx_mod.ops.gauss_fill(full, 3, 0, 0, cdx_mod.rng.STREAM_XT)
    full = full[..., :3].permute(0, 3, 1, 2)
    ct = T // 16
    tiles = torch.empty(1, len(ys), len(xs), 3, T, T)
    for iy, y in enumerate(ys):
        for ix, x in enumerate(xs):
            c = cond[:, :, y // 16:y // 16 + ct, x // 16:x // 16 + ct].contiguous()
            tiles[0, iy, ix] = sampler.sample(c, run["steps"], seed=0, x_T=full[:, :, y:y + T, x:x + T].contiguous())[0].cpu()
    want = oracle.blend_ref(tiles, ys, xs, S, S)
    got = out.cpu()
    err = (got - want).abs().max().item()
    # tile cores (pixels covered by exactly one tile) are copied, not blended: bit-exact
    cover = torch.zeros(S, S, dtype=torch.int32)
    for y in ys:
        for x in xs:
            cover[y:y + T, x:x + T] += 1
    core = cover == 1
    assert core.any() and torch.equal(got[0][:, core], want[0][:, core])
    # seams: inside every overlap the blend is a convex combination of the tiles that cover the pixel
    lo = torch.full((3, S, S), float("inf"))
    hi = torch.full((3, S, S), float("-inf"))
    for iy, y in enumerate(ys):
        for ix, x in enumerate(xs):
            lo[:, y:y + T, x:x + T] = torch.minimum(lo[:, y:y + T, x:x + T], tiles[0, iy, ix])
            hi[:, y:y + T, x:x + T] = torch.maximum(hi[:, y:y + T, x:x + T], tiles[0, iy, ix])
    assert (got[0] >= lo - 1e-6).all() and (got[0] <= hi + 1e-6).all()
    record("cfg5_full_1024_tiled", tiles=len(ys) * len(xs), max_err_vs_oracle_blend=err)
    assert err <= 2e-6
    # fp16 storage vs the float32 HIP path on the same image (5 steps): the stated fp16 tolerance
    tgt = torch.from_numpy(big["target"])
    cfg32 = dict(cfg16, dtype="fp32")
    a = sampler.sample_tiled(cond, 5, overlap=ov, seed=0).cpu()
    b = cdx_mod.Sampler(cdx_mod.UNet(cfg32, params)).sample_tiled(cond, 5, overlap=ov, seed=0).cpu()
    record("cfg5_full_1024_fp16_vs_fp32", psnr=psnr(a, b), dpsnr=abs(psnr(a, tgt) - psnr(b, tgt)))
    assert psnr(a, b) >= 45.0 and abs(psnr(a, tgt) - psnr(b, tgt)) <= 0.01
