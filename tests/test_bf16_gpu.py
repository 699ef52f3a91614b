"""bfloat16-storage path (dtype "bf16": SURVEY.md 8f rank 1 "fp16/bf16 storage path"): v_mfma_f32_32x32x16_bf16 kernels
against float64 statements evaluated on the SAME bf16-rounded operands, and the bf16 UNet / sampler against the float32
oracle with the tolerance stated after measurement (bf16 keeps 8 significand bits where fp16 keeps 11: errors are ~8x the
fp16 path's).  Needs a GPU: run with -m gpu."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cdx_mod(lib):
    import cdx
    assert torch.cuda.is_available()
    return cdx


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).float()


def b(t):       # round to bf16, back to float64 (the value the kernel actually sees)
    return t.bfloat16().double()


def nhwc16(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda().bfloat16()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu().double()


def psnr(x, y):
    mse = ((x.double() - y.double()) ** 2).mean().item()
    return float("inf") if mse == 0 else 10.0 * math.log10(4.0 / mse)


def test_bf16_weight_packer_rounds_to_nearest_even(cdx_mod):
    w = rnd(40, 64, 3, 3, seed=1, scale=0.05)
    img = cdx_mod._abi.pack_conv_weights_bf16(w.numpy(), 64, 0)
    got = torch.from_numpy(img[:-8192].view(np.int16)).view(torch.bfloat16).float().reshape(2, 2, 9, 2, 64, 8)
    want = w.bfloat16().float()
    for nt, ch, tap, j, lane, e in [(0, 0, 0, 0, 0, 0), (1, 1, 8, 1, 37, 5), (0, 1, 4, 1, 63, 7), (1, 0, 3, 0, 7, 2)]:
        n, c = nt * 32 + (lane & 31), ch * 32 + 16 * j + 8 * (lane >> 5) + e
        assert got[nt, ch, tap, j, lane, e].item() == (want[n, c].reshape(9)[tap].item() if n < 40 else 0.0)


@pytest.mark.parametrize("case", [(2, 32, 128, 32, 32, 3, 1, False), (1, 64, 160, 16, 16, 3, 1, False), (2, 64, 96, 32, 32, 1, 1, False),
                                  (2, 32, 128, 32, 32, 3, 2, False), (2, 32, 128, 16, 16, 3, 1, True), (1, 96, 128, 64, 64, 3, 1, False),
                                  (2, 160, 64, 8, 8, 3, 1, False), (3, 64, 100, 10, 12, 1, 1, False), (1, 256, 36, 9, 9, 3, 1, False)],      # 8 px wide: chunk-parallel tile
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_bf16_plain(cdx_mod, case):
    B, ci, co, H, W, k, s, up = case
    x = rnd(B, ci, H, W, seed=1)
    w = rnd(co, ci, k, k, seed=2, scale=1.0 / math.sqrt(ci * k * k))
    bias = rnd(co, seed=3)
    xin = F.interpolate(b(x), scale_factor=2, mode="nearest") if up else b(x)
    want = F.conv2d(xin, b(w), bias.double(), stride=s, padding=k // 2)
    pc = cdx_mod.ops.PackedConv16(w.numpy(), bias.numpy(), ci, bf16=True)
    got32 = nchw(cdx_mod.ops.conv16(pc, nhwc16(x), stride=s, upsample=up, out_dtype=torch.float32))
    assert (got32 - want).abs().max().item() <= 3e-6 * want.abs().max().item()          # fp32 accumulate of exact bf16 products
    got16 = nchw(cdx_mod.ops.conv16(pc, nhwc16(x), stride=s, upsample=up))
    assert got16.dtype == torch.float64 and (got16 - want).abs().max().item() <= 4.5e-3 * want.abs().max().item()   # + one rounding to bf16 (2^-9)


def test_conv_bf16_fused_gn_silu_concat_temb_residual_stats(cdx_mod):
    ops = cdx_mod.ops
    B, c0, c1, co, H, W, G = 2, 64, 32, 128, 32, 32, 32
    x0, x1 = rnd(B, c0, H, W, seed=4) * 2.0 + 0.7, rnd(B, c1, H, W, seed=5) * 0.5 - 1.0
    ci = c0 + c1
    gamma, beta = 1 + 0.2 * rnd(ci, seed=6), 0.3 * rnd(ci, seed=7)
    w = rnd(co, ci, 3, 3, seed=8, scale=1.0 / math.sqrt(ci * 9))
    bias, temb, res = rnd(co, seed=9), rnd(B, co + 5, seed=10), rnd(B, co, H, W, seed=11)
    xc = torch.cat([b(x0), b(x1)], 1)
    act = b(F.silu(F.group_norm(xc, G, gamma.double(), beta.double(), eps=1e-5)).float())   # the kernel rounds the activated input to bf16 in LDS
    want = F.conv2d(act, b(w), bias.double(), padding=1) + temb[:, 2:2 + co].double()[:, :, None, None] + b(res)
    s0, s1 = nhwc16(x0), nhwc16(x1)
    sc, sh = ops.gn_stats(s0.float().contiguous(), s1.float().contiguous(), gamma.cuda(), beta.cuda(), G)
    pc = ops.PackedConv16(w.numpy(), bias.numpy(), c0, c1, bf16=True)
    out, st = ops.conv16(pc, s0, s1, gn=(sc, sh), silu=True, temb=temb.cuda(), temb_off=2, residual=nhwc16(res), want_stats=True)
    assert out.dtype == torch.bfloat16
    assert (nchw(out) - want).abs().max().item() <= 2e-2 * want.abs().max().item()
    _, _, m2, r2 = ops.gn_finalize(st, None, H * W, torch.ones(co).cuda(), torch.zeros(co).cuda(), G, want_moments=True)
    wg = want.reshape(B, G, -1)
    assert (m2.cpu().double() - wg.mean(-1)).abs().max().item() <= 1.5e-2
    assert torch.allclose(r2.cpu().double(), (wg.var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=4e-2)


def test_conv_bf16_chunk_parallel_tile_fused_with_sums(cdx_mod):
    """8 x 8 level in bf16 storage (KparCfg<..., SPLIT = 0, BF = 1>): GroupNorm + SiLU on load, temb, residual and the four
    GroupNorm-sum slots per tile, 5 input chunks over 4 waves (uneven split), ragged 10 x 9 image."""
    ops = cdx_mod.ops
    B, ci, co, H, W, G = 2, 160, 64, 10, 9, 32
    x = rnd(B, ci, H, W, seed=40) * 1.5 + 0.3
    gamma, beta = 1 + 0.2 * rnd(ci, seed=41), 0.3 * rnd(ci, seed=42)
    w = rnd(co, ci, 3, 3, seed=43, scale=1.0 / math.sqrt(ci * 9))
    bias, temb, res = rnd(co, seed=44), rnd(B, co, seed=45), rnd(B, co, H, W, seed=46)
    act = b(F.silu(F.group_norm(b(x), G, gamma.double(), beta.double(), eps=1e-5)).float())
    want = F.conv2d(act, b(w), bias.double(), padding=1) + temb.double()[:, :, None, None] + b(res)
    s0 = nhwc16(x)
    sc, sh = ops.gn_stats(s0.float().contiguous(), None, gamma.cuda(), beta.cuda(), G)
    pc = ops.PackedConv16(w.numpy(), bias.numpy(), ci, bf16=True)
    out, st = ops.conv16(pc, s0, gn=(sc, sh), silu=True, temb=temb.cuda(), residual=nhwc16(res), want_stats=True)
    assert out.dtype == torch.bfloat16
    assert (nchw(out) - want).abs().max().item() <= 2e-2 * want.abs().max().item()
    _, _, m2, r2 = ops.gn_finalize(st, None, H * W, torch.ones(co).cuda(), torch.zeros(co).cuda(), G, want_moments=True)
    wg = want.reshape(B, G, -1)
    assert (m2.cpu().double() - wg.mean(-1)).abs().max().item() <= 1.5e-2
    assert torch.allclose(r2.cpu().double(), (wg.var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=4e-2)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 6e-3), (torch.float16, 8e-4)])
def test_attention_16bit_mfma(cdx_mod, dtype, tol):
    """Both contractions on the 16-bit matrix pipe, float32 softmax: ragged query / key counts, two heads."""
    B, heads, nq, nk, hd = 2, 2, 200, 77, 64
    C = heads * hd
    q, k, v = rnd(B, nq, C, seed=30), rnd(B, nk, C, seed=31), rnd(B, nk, C, seed=32)
    r = (lambda t: t.to(dtype).double())
    qh, kh, vh = (r(t).reshape(B, -1, heads, hd).transpose(1, 2) for t in (q, k, v))
    want = (torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, -1) @ vh).transpose(1, 2).reshape(B, nq, C)
    got = cdx_mod.ops.attention(q.cuda().to(dtype), k.cuda().to(dtype), v.cuda().to(dtype), heads)
    assert got.dtype == dtype
    assert (got.cpu().double() - want).abs().max().item() <= tol * want.abs().max().item()


@pytest.mark.parametrize("name,over", [
    ("tiny", dict(image_size=16, base_channels=32, channel_mult=(1, 2), attn_resolutions=(8,), num_res_blocks=1)),
    ("wide", dict(image_size=64, base_channels=128, channel_mult=(1, 2, 2), attn_resolutions=(16,), num_res_blocks=1)),
    ("xattn", dict(image_size=64, base_channels=64, channel_mult=(1, 2), cond_mode="cross_attn", attn_resolutions=(32,),
                   cross_attn_resolutions=(64, 32), context_dim=96, num_res_blocks=1)),
])
def test_bf16_unet_forward_vs_fp32_oracle(cdx_mod, record, name, over):
    """bf16 storage against the float64 oracle: stated tolerance 3e-2 of the output scale (measured 1.0-1.4e-2; fp16: 1e-2 stated, 1.5e-3 measured)."""
    import oracle
    cfg32 = cdx_mod.unet_config(**over)
    cfg16 = cdx_mod.unet_config(**over, dtype="bf16")
    params = cdx_mod.init_params(cfg32, seed=2, affine_jitter=0.1, out_gain=1.0)
    B = 2
    cond = torch.from_numpy(cdx_mod.synthetic_batch(cfg32, 2, 0, B)["cond"])
    x = torch.randn(B, 3, cfg32["image_size"], cfg32["image_size"], generator=torch.Generator().manual_seed(1))
    t = torch.tensor([900, 40])
    want = oracle.unet_forward_ref(cfg32, params, x, t, cond, dtype=torch.float64)
    got = cdx_mod.UNet(cfg16, params).forward(x.cuda(), t.cuda(), cond.cuda()).cpu()
    err = (got.double() - want).abs().max().item() / want.abs().max().item()
    record("bf16_unet_forward_" + name, rel_max_err=err, rms_rel=((got.double() - want).pow(2).mean().sqrt() / want.abs().max()).item())
    assert err <= 3e-2


def test_bf16_sampler_vs_fp32_oracle(cdx_mod, record):
    """cfg1-shaped 50-step DDIM in bf16 storage vs the committed float32 oracle output.  The 0.01 dB gate is a float32
    statement; bf16 tolerance stated after measurement: PSNR(bf16, oracle) >= 28 dB, |dPSNR vs target| <= 0.05 dB."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1_ddim50.npz"))
    cfg32, run = cdx_mod.named_config("cfg1")
    cfg16 = dict(cfg32, dtype="bf16")
    params = cdx_mod.init_params(cfg32, seed=0)
    sb = cdx_mod.synthetic_batch(cfg32, 0, 0, 1)
    got = cdx_mod.Sampler(cdx_mod.UNet(cfg16, params)).sample(torch.from_numpy(sb["cond"]).cuda(), run["steps"], seed=0).cpu()
    want, tgt = torch.from_numpy(g["x0"]), torch.from_numpy(sb["target"])
    record("bf16_sampler_cfg1", psnr_bf16_vs_oracle=psnr(got, want), dpsnr=abs(psnr(got, tgt) - psnr(want, tgt)))
    assert psnr(got, want) >= 28.0 and abs(psnr(got, tgt) - psnr(want, tgt)) <= 0.05
