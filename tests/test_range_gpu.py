"""Range contract of the float32 convolution on the fp16 matrix pipe (include/cdx.h "RANGE CONTRACT"):
the split-fp16 tiles must reproduce F.conv2d at float32-level error relative to the OUTPUT scale at ANY input scale, let
NaN / Inf through as F.conv2d does, and never overflow on their own scaling.  Every case runs against float64 torch at
the SAME tolerances as the unit-scale kernel tests (2e-6 plain, 3e-6 fused).  Needs a GPU: -m gpu."""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SPLIT = 11
SCALES = [1e-6, 1e-3, 1.0, 1e3, 6e4, 1e6]


@pytest.fixture(scope="module")
def ops(lib):
    import cdx
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return cdx.ops


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).float()


def rel_err(got, want):
    want = want.to(torch.float64)
    return (got.to(torch.float64) - want).abs().max().item() / max(want.abs().max().item(), 1e-300)


def takes_split(ops, pc, x, x1=None, **kw):
    import cdx
    out = torch.empty(1, device="cuda")      # placeholder: only the shape fields are read
    B, h, w, _ = x.shape
    up, s = kw.get("upsample", False), kw.get("stride", 1)
    hv, wv = (2 * h, 2 * w) if up else (h, w)
    o = torch.empty(B, (hv + s - 1) // s, (wv + s - 1) // s, pc.cout, device="cuda")
    am = ops.amax_buffer(B, "cuda")
    a = ops.conv_args(pc, x, x1, o, src_amax=(am, am), **kw)
    return cdx._abi.lib().cdx_conv_select_tile(ctypes.byref(a)) == SPLIT


# (B, Cin, Cout, H, W, ksize, stride, upsample): one case per split tile shape
SHAPES = [
    (2, 64, 128, 32, 32, 3, 1, False),     # 128 px x 128 ch tile
    (1, 96, 192, 40, 36, 3, 1, False),     # ragged, 128 + 64 channels (2 x 2 tail layout)
    (2, 128, 96, 16, 16, 3, 1, False),     # 64 x 128 tile at 16 px
    (3, 256, 64, 8, 8, 3, 1, False),       # chunk-parallel tile at 8 px
    (2, 64, 128, 36, 34, 3, 2, False),  # stride 2 (one halo image)
    (2, 64, 96, 32, 32, 1, 1, False),      # 1x1
    (2, 96, 160, 8, 8, 1, 1, False),       # 1x1 chunk-parallel
    (2, 64, 64, 8, 8, 3, 1, True),         # nearest-2x upsample fused, 16 px out
    (1, 64, 128, 36, 32, 3, 1, True),      # nearest-2x upsample as four 2x2 phase convolutions (low-resolution rows >= 32 px)
]


@pytest.mark.parametrize("scale", SCALES, ids=lambda s: f"x{s:g}")
@pytest.mark.parametrize("shape", SHAPES, ids=lambda c: "x".join(map(str, c)))
def test_conv_any_activation_scale(ops, record, shape, scale):
    """x ~ N(0, scale^2): split tile (forced and as the library's own pick) vs float64 F.conv2d, tolerance unchanged."""
    B, ci, co, H, W, k, s, up = shape
    x = rnd(B, ci, H, W, seed=1, scale=scale)
    w = rnd(co, ci, k, k, seed=2, scale=1.0 / math.sqrt(ci * k * k))
    bias = rnd(co, seed=3, scale=scale)                   # additive terms at the output's scale
    xin = F.interpolate(x.double(), scale_factor=2, mode="nearest") if up else x.double()
    want = F.conv2d(xin, w.double(), bias.double(), stride=s, padding=k // 2)
    pc = ops.PackedConv(w.numpy(), bias.numpy(), ci)
    xd = nhwc(x)
    assert takes_split(ops, pc, xd, stride=s, upsample=up)
    for tile in (SPLIT, -1):
        got = nchw(ops.conv(pc, xd, stride=s, upsample=up, tile=tile))
        assert torch.isfinite(got).all()
        e = rel_err(got, want)
        record("range_conv_scale", shape="x".join(map(str, shape)), scale=scale, tile=tile, rel_err=e)
        assert e <= 2e-6, f"tile {tile} scale {scale:g}: rel err {e:.3e}"


@pytest.mark.parametrize("shape", SHAPES[:4] + SHAPES[5:6], ids=lambda c: "x".join(map(str, c)))
def test_conv_mixed_scale_channels_and_images(ops, record, shape):
    """Channel c at scale 10^U(-3, 3), image b at another 10^(3 b): the per-IMAGE exponent keeps every image at float32-level
    error relative to its own output scale (a per-launch exponent would not)."""
    B, ci, co, H, W, k, s, up = shape
    g = torch.Generator().manual_seed(7)
    cs = 10.0 ** (6 * torch.rand(ci, generator=g, dtype=torch.float64) - 3)
    bs = 10.0 ** (3.0 * torch.arange(B, dtype=torch.float64) - 3)
    x = (rnd(B, ci, H, W, seed=8).double() * cs[None, :, None, None] * bs[:, None, None, None]).float()
    w = rnd(co, ci, k, k, seed=9, scale=1.0 / math.sqrt(ci * k * k))
    want = F.conv2d(x.double(), w.double(), None, stride=s, padding=k // 2)
    pc = ops.PackedConv(w.numpy(), None, ci)
    got = nchw(ops.conv(pc, nhwc(x), stride=s, tile=SPLIT))
    for b in range(B):
        e = rel_err(got[b], want[b])
        record("range_conv_mixed", shape="x".join(map(str, shape)), image=b, rel_err=e)
        assert e <= 2e-6, f"image {b}: rel err {e:.3e}"


def test_conv_concat_sources_at_different_scales(ops, record):
    """cat[x (1e3), skip (1e-2)]: one exponent per image from the larger source; error relative to the output scale."""
    B, c0, c1, co, H, W = 2, 64, 32, 128, 32, 32
    x0, x1 = rnd(B, c0, H, W, seed=11, scale=1e3), rnd(B, c1, H, W, seed=12, scale=1e-2)
    w = rnd(co, c0 + c1, 3, 3, seed=13, scale=1.0 / math.sqrt((c0 + c1) * 9))
    want = F.conv2d(torch.cat([x0, x1], 1).double(), w.double(), None, padding=1)
    pc = ops.PackedConv(w.numpy(), None, c0, c1)
    got = nchw(ops.conv(pc, nhwc(x0), nhwc(x1), tile=SPLIT))
    e = rel_err(got, want)
    record("range_conv_concat", rel_err=e)
    assert e <= 2e-6


@pytest.mark.parametrize("gscale", [1e-4, 1.0, 300.0], ids=lambda s: f"gamma{s:g}")
@pytest.mark.parametrize("xscale", [1e-5, 1.0, 1e5], ids=lambda s: f"x{s:g}")
@pytest.mark.parametrize("shape", [(2, 64, 128, 32, 32), (2, 128, 64, 8, 8), (1, 64, 96, 16, 16)], ids=lambda c: "x".join(map(str, c)))
def test_fused_gn_silu_conv_any_gamma_and_input_scale(ops, record, shape, xscale, gscale):
    """conv3x3(silu(gn(x))) + temb + residual with gamma / beta far from 1: the static exponent (cdx_gn_act_exp) keeps the staged
    tensor inside fp16's range and precision at the unchanged fused tolerance."""
    B, ci, co, H, W = shape
    x = rnd(B, ci, H, W, seed=21, scale=xscale) + 0.7 * xscale
    gamma, beta = gscale * (1 + 0.2 * rnd(ci, seed=22)), gscale * 0.3 * rnd(ci, seed=23)
    w = rnd(co, ci, 3, 3, seed=24, scale=1.0 / math.sqrt(ci * 9))
    osc = gscale                                              # output scale follows gamma (SiLU is ~linear for |v| >> 1 and ~v/2 near 0)
    bias, temb, res = rnd(co, seed=25, scale=osc), rnd(B, co, seed=26, scale=osc), rnd(B, co, H, W, seed=27, scale=osc)
    # eps = 1e-5 is part of the definition: at xscale 1e-5 it dominates the variance -- the oracle sees the same float32 x
    h = F.silu(F.group_norm(x.double(), 32, gamma.double(), beta.double(), eps=1e-5))
    want = F.conv2d(h, w.double(), bias.double(), padding=1) + temb.double()[:, :, None, None] + res.double()
    pc = ops.PackedConv(w.numpy(), bias.numpy(), ci)
    got = nchw(ops.conv(pc, nhwc(x), gn_affine=(gamma.cuda(), beta.cuda(), 32), silu=True, temb=temb.cuda(), residual=nhwc(res), tile=SPLIT))
    assert torch.isfinite(got).all()
    e = rel_err(got, want)
    record("range_fused_gn", shape="x".join(map(str, shape)), xscale=xscale, gscale=gscale, rel_err=e)
    assert e <= 3e-6, f"rel err {e:.3e}"


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), -float("inf")], ids=["nan", "inf", "-inf"])
@pytest.mark.parametrize("shape", SHAPES[:1] + SHAPES[3:6], ids=lambda c: "x".join(map(str, c)))
def test_nonfinite_inputs_propagate_like_conv2d(ops, shape, bad):
    """One NaN / Inf input element: the outputs F.conv2d makes non-finite are non-finite, every other output is finite and as
    accurate as without it (other pixels of the image, and the other images)."""
    B, ci, co, H, W, k, s, up = shape
    x = rnd(B, ci, H, W, seed=31)
    x[0, 5, H // 2, W // 3] = bad
    w = rnd(co, ci, k, k, seed=32, scale=1.0 / math.sqrt(ci * k * k))
    want = F.conv2d(x.double(), w.double(), None, stride=s, padding=k // 2)
    pc = ops.PackedConv(w.numpy(), None, ci)
    for tile in (SPLIT, -1):
        got = nchw(ops.conv(pc, nhwc(x), stride=s, tile=tile))
        wf = torch.isfinite(want)
        assert (~wf).any() and torch.equal(torch.isfinite(got), wf), f"tile {tile}: non-finite footprint differs"
        g0, w0 = torch.where(wf, got.double(), torch.zeros((), dtype=torch.float64)), torch.where(wf, want, torch.zeros((), dtype=torch.float64))
        assert rel_err(g0, w0) <= 2e-6
    with pytest.raises(Exception, match="non-finite"):
        ops.conv(pc, nhwc(x), stride=s, tile=SPLIT, debug=True)


def test_nonfinite_through_groupnorm_poisons_only_its_image(ops):
    B, ci, co, H, W = 3, 64, 128, 32, 32
    x = rnd(B, ci, H, W, seed=41)
    x[1, 3, 4, 5] = float("nan")
    gamma, beta = torch.ones(ci), torch.zeros(ci)
    w = rnd(co, ci, 3, 3, seed=42, scale=1.0 / math.sqrt(ci * 9))
    want = F.conv2d(F.silu(F.group_norm(x.double(), 32, gamma.double(), beta.double())), w.double(), None, padding=1)
    pc = ops.PackedConv(w.numpy(), None, ci)
    got = nchw(ops.conv(pc, nhwc(x), gn_affine=(gamma.cuda(), beta.cuda(), 32), silu=True, tile=SPLIT))
    assert torch.equal(torch.isfinite(got), torch.isfinite(want))
    assert not torch.isfinite(got[1]).any() and rel_err(got[[0, 2]], want[[0, 2]]) <= 3e-6


@pytest.mark.parametrize("wscale,xscale,addscale", [
    (1e-30, 1.0, 1.0),       # near-zero weights, O(1) bias / temb / residual: out = the additive terms
    (0.0, 1.0, 1.0),         # all-zero weights
    (1e-12, 1e-12, 1.0),     # 2^S beyond the accumulator-init range: additive terms applied in the epilogue
    (1e-12, 1e-12, 0.0),     # ... and nothing added: the tiny products themselves, relative to THEIR scale
    (1e-15, 1e-6, 1e-21),    # additive terms at the products' scale
    (1e10, 1e8, 1e18),       # huge x huge
    (1.0, 1.0, 1e15),        # huge residual (additive terms are safe below 2^63: cdx.h)
])
@pytest.mark.parametrize("shape", [(2, 64, 128, 32, 32), (2, 128, 64, 8, 8)], ids=["tile128", "kpar"])
def test_extreme_weight_and_additive_scales(ops, record, shape, wscale, xscale, addscale):
    B, ci, co, H, W = shape
    x = rnd(B, ci, H, W, seed=51, scale=xscale)
    w = rnd(co, ci, 3, 3, seed=52, scale=wscale / math.sqrt(ci * 9))
    bias, temb, res = rnd(co, seed=53, scale=addscale), rnd(B, co, seed=54, scale=addscale), rnd(B, co, H, W, seed=55, scale=addscale)
    want = F.conv2d(x.double(), w.double(), bias.double(), padding=1) + temb.double()[:, :, None, None] + res.double()
    pc = ops.PackedConv(w.numpy(), bias.numpy(), ci)
    got = nchw(ops.conv(pc, nhwc(x), temb=temb.cuda(), residual=nhwc(res), tile=SPLIT))
    assert torch.isfinite(got).all()
    e = rel_err(got, want)
    record("range_extreme", shape="x".join(map(str, shape)), wscale=wscale, xscale=xscale, addscale=addscale, rel_err=e)
    assert e <= 3e-6, f"rel err {e:.3e}"


@pytest.mark.parametrize("shape,tile", [((2, 64, 128, 32, 32, 3, 1), SPLIT), ((2, 128, 64, 8, 8, 3, 1), SPLIT), ((2, 64, 128, 36, 34, 3, 2), SPLIT),
                                        ((2, 64, 96, 40, 24, 1, 1), SPLIT), ((2, 32, 64, 4, 4, 3, 1), -1), ((1, 64, 128, 32, 32, 3, 1), 7), ((1, 64, 128, 32, 32, 3, 1), 0)],
                         ids=["tile128", "kpar", "stride2", "k1", "splitK4x4", "wino", "direct"])
def test_amax_out_is_the_per_image_maximum(ops, shape, tile):
    """amax_out (every tile shape): bit pattern of max |out[b]|, max-combined into the caller's word; stats entries all written."""
    B, ci, co, H, W, k, s = shape
    x = rnd(B, ci, H, W, seed=61) * torch.tensor([1.0, 37.0][:B])[:, None, None, None]
    w = rnd(co, ci, k, k, seed=62, scale=1.0 / math.sqrt(ci * k * k))
    pc = ops.PackedConv(w.numpy(), rnd(co, seed=63).numpy(), ci)
    if tile in (SPLIT, -1):                               # (sums are defined for the library's own tile choice only)
        out, stats, am = ops.conv(pc, nhwc(x), stride=s, tile=tile, want_stats=True, want_amax=True)
        assert not torch.isnan(stats).any()
    else:
        out, am = ops.conv(pc, nhwc(x), stride=s, tile=tile, want_amax=True)
    want = out.abs().amax(dim=(1, 2, 3))
    assert torch.equal(ops.amax_value(am), want), (ops.amax_value(am), want)
    # standalone pass (tensors no convolution produced): the same maxima
    assert torch.equal(ops.amax_value(ops.amax(out)), want)


def test_unnormalised_launch_without_amax_falls_back_to_f32_tiles(ops):
    """cdx.h: an un-normalised launch that brings no src_amax never takes the split tile (and still computes the right thing)."""
    import cdx
    B, ci, co, H, W = 1, 64, 128, 32, 32
    x = rnd(B, ci, H, W, seed=71, scale=1e-4)
    w = rnd(co, ci, 3, 3, seed=72, scale=1.0 / math.sqrt(ci * 9))
    pc = ops.PackedConv(w.numpy(), None, ci)
    xd = nhwc(x)
    out = torch.empty(B, H, W, co, device="cuda")
    a = ops.conv_args(pc, xd, None, out)
    assert cdx._abi.lib().cdx_conv_select_tile(ctypes.byref(a)) != SPLIT
    assert cdx._abi.lib().cdx_conv_f32_tile(ctypes.byref(a), SPLIT, None, 0, None) == -4          # CDX_ENOTSUP
    got = nchw(ops.conv(pc, xd, auto_range=False))
    assert rel_err(got, F.conv2d(x.double(), w.double(), None, padding=1)) <= 4e-6
    # a GroupNorm exponent on a launch that does not take the split tile is an error, not a silent mis-scale
    sc, sh = torch.ones(B, ci, device="cuda"), torch.zeros(B, ci, device="cuda")
    a = ops.conv_args(pc, xd, None, out, gn=(sc, sh, 5))
    assert cdx._abi.lib().cdx_conv_f32_tile(ctypes.byref(a), 0, None, 0, None) == -1
    # ADVICE r03 (ABI v5): a GroupNorm-ed launch that does NOT state its exponent (the plain pair of ops.gn_stats) never takes the
    # split tile -- unit-scale fp16 staging without a clamp would turn a large |gamma| x_hat into Inf where the f32 tiles are exact
    a = ops.conv_args(pc, xd, None, out, gn=(sc, sh))
    assert not (a.flags & cdx._abi.CONV_GN_EXP) and a.gn_exp == 0
    assert cdx._abi.lib().cdx_conv_select_tile(ctypes.byref(a)) != SPLIT
    assert cdx._abi.lib().cdx_conv_f32_tile(ctypes.byref(a), SPLIT, None, 0, None) == -4          # CDX_ENOTSUP
    a.gn_exp = 3                                                                                   # an exponent without the flag: EINVAL
    assert cdx._abi.lib().cdx_conv_f32_tile(ctypes.byref(a), -1, None, 0, None) == -1
    a = ops.conv_args(pc, xd, None, out, gn=(sc, sh, 0))                                           # stated (even 0): the split tile
    assert (a.flags & cdx._abi.CONV_GN_EXP) and cdx._abi.lib().cdx_conv_select_tile(ctypes.byref(a)) == SPLIT
    gam = torch.full((ci,), 300.0)                  # |gamma| 300: 300 x 4-sigma values are beyond fp16's 65504 at unit scale...
    xb = rnd(B, ci, H, W, seed=73) ** 3             # (heavy tails: |x_hat| up to ~30)
    want = F.conv2d(F.group_norm(xb.double(), 32, gam.double(), torch.zeros(ci).double()), w.double(), None, padding=1)
    xbd = nhwc(xb)
    plain = nchw(ops.conv(pc, xbd, gn=ops.gn_stats(xbd, None, gam.cuda(), torch.zeros(ci).cuda(), 32)))      # ... so the plain pair runs on f32 tiles
    stated = nchw(ops.conv(pc, xbd, gn_affine=(gam.cuda(), torch.zeros(ci).cuda(), 32)))                    # ... and the stated form scales it down
    assert torch.isfinite(plain).all() and torch.isfinite(stated).all()
    assert rel_err(plain, want) <= 4e-6 and rel_err(stated, want) <= 4e-6


def test_sources_beyond_4_gib_index_correctly(ops):
    """32-bit byte offsets are taken inside ONE image (per-image buffer resource): a 4.3 GB source batch decodes its
    last image to the same bits as that image alone."""
    B, ci, co, H, W = 129, 128, 32, 256, 256
    assert B * H * W * ci * 4 > 2 ** 32
    x = torch.empty(B, H, W, ci, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    x.normal_(generator=g)
    w = rnd(co, ci, 3, 3, seed=82, scale=1.0 / math.sqrt(ci * 9))
    pc = ops.PackedConv(w.numpy(), None, ci)
    got = ops.conv(pc, x, tile=SPLIT)
    for b in (0, 127, 128):
        one = ops.conv(pc, x[b:b + 1].contiguous(), tile=SPLIT)
        assert torch.equal(got[b:b + 1], one), f"image {b}"
    want = F.conv2d(x[128:129].permute(0, 3, 1, 2).double().cpu(), w.double(), None, padding=1)
    assert rel_err(nchw(got[128:129]), want) <= 2e-6


def test_unet_forward_debug_mode_names_the_layer(lib):
    """Plan.run(debug=True): a NaN planted in the input makes the first convolution that stores it raise, by name."""
    import cdx
    cfg = cdx.unet_config(image_size=16, base_channels=32, channel_mult=(1, 2), attn_resolutions=(8,), num_res_blocks=1)
    net = cdx.UNet(cfg, cdx.init_params(cfg, seed=3), device="cuda:0")
    p = net.plan(2)
    p.xin.normal_()
    p.run(debug=True)                                    # clean input: no complaint
    torch.cuda.synchronize()
    p.xin[1, 3, 3, 0] = float("nan")
    with pytest.raises(cdx._abi.CdxError, match="conv_in"):
        p.run(debug=True)
    p.run()
    torch.cuda.synchronize()
    assert torch.isfinite(p.eps[0]).all() and not torch.isfinite(p.eps[1, ..., :3]).any()


def test_unet_forward_at_tiny_and_huge_input_scale(lib, record):
    """Whole UNet forward, float64 oracle, inputs and cond at 1e-4 / 1e4 of the usual scale: the un-normalised launches (conv_in,
    skips, down / up convs, attention projection) take their exponents from the producers' maxima."""
    import cdx
    import oracle
    cfg = cdx.unet_config(image_size=32, base_channels=32, channel_mult=(1, 2, 2), attn_resolutions=(16,), num_res_blocks=1)
    params = cdx.init_params(cfg, seed=5, affine_jitter=0.1)
    net = cdx.UNet(cfg, params, device="cuda:0")
    net_f32 = cdx.UNet(cfg, params, device="cuda:0", split=False)
    t = torch.tensor([500, 17])
    for scale in (1e-4, 1.0, 1e4):
        x = rnd(2, 3, 32, 32, seed=91, scale=scale)
        cond = rnd(2, 3, 2, 2, seed=92, scale=scale)
        want = oracle.unet_forward_ref(cfg, params, x.double(), t, cond.double(), dtype=torch.float64)
        e32 = rel_err(oracle.unet_forward_ref(cfg, params, x, t, cond), want)        # the CPU float32 oracle's own distance
        got = net(x.cuda(), t.cuda(), cond.cuda()).cpu()
        e = rel_err(got, want)
        e_f32 = rel_err(net_f32(x.cuda(), t.cuda(), cond.cuda()).cpu(), want)      # the f32-input MFMA kernels at the same scale
        record("range_unet_forward", scale=scale, rel_err=e, rel_err_f32mfma=e_f32, cpu_fp32_rel_err=e32)
        assert e_f32 <= max(4e-6, 2 * e32), f"scale {scale:g} (split=False): rel err {e_f32:.3e} (CPU float32 oracle: {e32:.3e})"
        # At 1e-4 the first GroupNorm sees conv_in's bias (0.1) plus a 1e-4 signal -- 2000 sigmas from zero: the float32 forward
        # itself is accurate to ~1e-4 of the signal there (e32).  Round 3 sat at 5x that and blamed the split operands; the cause
        # was the additive terms riding in the accumulators (re-rounded at the bias' ulp by every MFMA) and the float32 shift
        # of the GroupNorm (profiles/r04_parity_metrics.jsonl: per-layer error per variant, tools/diag_scale.py).  Bias + temb
        # now enter in the epilogue's FMA and the shift is evaluated in float64: the gate is back at 2x the CPU float32 oracle's
        # own distance from float64 (4e-6 where the problem is well conditioned).
        assert e <= max(4e-6, 2 * e32), f"scale {scale:g}: rel err {e:.3e} (CPU float32 oracle: {e32:.3e})"
