"""Deterministic boundary classes of the convolution tiles: images LOWER than a tile (H = 1, 2, 3:
whole tile rows -- for the 2 x 2 tail layout whole WAVES -- lie past the image), channel counts that leave a partial last block
(129, 160 = 128 + 32, 192 = 128 + 64, 320 = 2 x 128 + 64), the last image of the batch (whatever is read or written past it
lies past the tensor), for every tile geometry the library ships: the 8 x 16 wave-specialised tile, the 64-pixel wave-specialised
tile, the chunk-parallel tile, the 4 x 32 1x1 tile, both stride-2 tiles, the four-phase and the gather-fused up-sampling forms --
plain and with GroupNorm + SiLU + temb + residual + GroupNorm sums, in float32 (every tile id that accepts the launch), fp16 and
bf16.  Round 3's 800-case fuzz met one member of this class as a GPU memory fault; a random sweep gives it ~1 % weight, this
file enumerates it.  Outputs are views in front of a guard region that must survive.  Needs a GPU: run with -m gpu."""
import math

import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import close, nchw, nhwc, rnd, tiles_for

pytestmark = pytest.mark.gpu

GEOMS = {          # name: (W, ksize, stride, upsample)
    "k3_w40": (40, 3, 1, False),       # 8 x 16-pixel wave-specialised tile, ragged in x
    "k3_w16": (16, 3, 1, False),       # 64-pixel wave-specialised tile
    "k3_w8": (8, 3, 1, False),         # chunk-parallel tile
    "k1_w40": (40, 1, 1, False),       # 1x1: 4 x 32-pixel 4-wave tile
    "k1_w8": (8, 1, 1, False),         # 1x1 chunk-parallel
    "k3s2_w66": (66, 3, 2, False),     # stride 2, 33 wide out
    "k3s2_w32": (32, 3, 2, False),     # stride 2, 16 wide out
    "up_w32": (32, 3, 1, True),        # four 2x2 phase convolutions on the low-resolution source (no residual) / gather form (residual)
    "up_w12": (12, 3, 1, True),        # gather-fused nearest-2x
}
GUARD = 4096
SENTINEL = 12345.0


@pytest.fixture(scope="module")
def cdx_mod(lib):
    import cdx
    assert torch.cuda.is_available()
    return cdx


def guarded(shape, dtype):
    """A tensor of `shape` that is the FRONT of a larger allocation whose tail holds a sentinel (a store past the tensor lands there)."""
    n = math.prod(shape)
    big = torch.full((n + GUARD,), SENTINEL, device="cuda", dtype=dtype)
    return big, big[:n].view(*shape)


def _case(H, geom, co, fused, seed=0):
    W, k, s, up = GEOMS[geom]
    B, ci, G = 2, 64, 32
    x = rnd(B, ci, H, W, seed=200 + seed) * 1.5 + 0.4
    w = rnd(co, ci, k, k, seed=201 + seed, scale=1.0 / math.sqrt(ci * k * k))
    bias = rnd(co, seed=202 + seed)
    gamma, beta = 1 + 0.2 * rnd(ci, seed=203 + seed), 0.3 * rnd(ci, seed=204 + seed)
    Hv, Wv = (2 * H, 2 * W) if up else (H, W)
    ho, wo = (Hv + s - 1) // s, (Wv + s - 1) // s
    temb, res = rnd(B, co + 2, seed=205 + seed), rnd(B, co, ho, wo, seed=206 + seed)
    return dict(B=B, ci=ci, G=G, x=x, w=w, bias=bias, gamma=gamma, beta=beta, temb=temb, res=res, k=k, s=s, up=up, ho=ho, wo=wo, W=W)


def _want(c, x, w, res, fused, act_round=None):
    h = x.double()
    if fused:
        h = F.silu(F.group_norm(h, c["G"], c["gamma"].double(), c["beta"].double(), eps=1e-5))
        if act_round is not None:
            h = h.to(act_round).double()
    if c["up"]:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
    y = F.conv2d(h, w.double(), c["bias"].double(), stride=c["s"], padding=c["k"] // 2)
    if fused:
        y = y + c["temb"][:, 1:1 + w.shape[0]].double()[:, :, None, None] + res.double()
    return y


@pytest.mark.parametrize("fused", [False, True], ids=["plain", "gn_silu_temb_res_stats"])
@pytest.mark.parametrize("co", [129, 160, 192, 320])
@pytest.mark.parametrize("geom", list(GEOMS))
@pytest.mark.parametrize("H", [1, 2, 3])
def test_f32_tiles_lower_than_a_tile(cdx_mod, H, geom, co, fused):
    ops = cdx_mod.ops
    c = _case(H, geom, co, fused)
    want = _want(c, c["x"], c["w"], c["res"], fused)
    pc = ops.PackedConv(c["w"].numpy(), c["bias"].numpy(), c["ci"], up=c["up"])
    xd = nhwc(c["x"])
    stats_ok = fused and co % 4 == 0
    for tile in tiles_for(c["k"], c["s"], c["wo"], co, c["ci"]):
        big, out = guarded((c["B"], c["ho"], c["wo"], co), torch.float32)
        out.fill_(float("nan"))
        kw = dict(stride=c["s"], upsample=c["up"], tile=tile, out=out)
        if fused:
            kw.update(gn_affine=(c["gamma"].cuda(), c["beta"].cuda(), c["G"]), silu=True, temb=c["temb"].cuda(), temb_off=1,
                      residual=nhwc(c["res"]), want_stats=stats_ok)
        try:
            r = ops.conv(pc, xd, **kw)
        except cdx_mod._abi.CdxError as e:       # a forced tile id may refuse a launch it is not built for -- by status, never by fault
            assert tile >= 0, f"the library's own pick failed: {e}"
            continue
        torch.cuda.synchronize()
        assert (big[out.numel():] == SENTINEL).all(), f"tile {tile}: wrote past the output tensor"
        close(nchw(out), want, 5e-6 if tile == 7 else 3e-6, f"H={H} {geom} cout={co} tile {tile}")
        if stats_ok:
            st = r[1]
            assert not torch.isnan(st).any(), f"tile {tile}: a GroupNorm-sum slot was left unwritten"
            sums = st[..., 0].sum(1).cpu()          # [B, slots, C, 2] -> per (image, channel)
            close(sums, nchw(out).double().sum((2, 3)), 1e-5, f"H={H} {geom} cout={co} tile {tile} sums")


@pytest.mark.parametrize("fused", [False, True], ids=["plain", "gn_silu_temb_res_stats"])
@pytest.mark.parametrize("co", [132, 160, 192, 320])      # (16-bit outputs move as 4-channel vectors: cdx_conv_f16 wants cout % 4 == 0 here)
@pytest.mark.parametrize("geom", list(GEOMS))
@pytest.mark.parametrize("H", [1, 2, 3])
@pytest.mark.parametrize("bf16", [False, True], ids=["fp16", "bf16"])
def test_16bit_tiles_lower_than_a_tile(cdx_mod, bf16, H, geom, co, fused):
    ops = cdx_mod.ops
    dt = torch.bfloat16 if bf16 else torch.float16
    c = _case(H, geom, co, fused, seed=50)
    rd = lambda t: t.to(dt).float()      # noqa: E731  (the values the kernel sees)
    want = _want(c, rd(c["x"]), rd(c["w"]), rd(c["res"]), fused, act_round=dt)
    pc = ops.PackedConv16(c["w"].numpy(), c["bias"].numpy(), c["ci"], bf16=bf16)
    xd = nhwc(c["x"]).to(dt)
    big, out = guarded((c["B"], c["ho"], c["wo"], co), dt)
    out.fill_(float("nan"))
    kw = dict(stride=c["s"], upsample=c["up"], out=out)
    if fused:
        sc, sh = ops.gn_stats(xd.float().contiguous(), None, c["gamma"].cuda(), c["beta"].cuda(), c["G"])
        kw.update(gn=(sc, sh), silu=True, temb=c["temb"].cuda(), temb_off=1, residual=nhwc(c["res"]).to(dt), want_stats=co % 4 == 0)
    r = ops.conv16(pc, xd, **kw)
    torch.cuda.synchronize()
    assert (big[out.numel():] == torch.tensor(SENTINEL).to(dt).item()).all(), "wrote past the output tensor"
    tol = (3e-2 if fused else 8e-3) if bf16 else (4e-3 if fused else 1e-3)
    close(nchw(out.float()), want, tol, f"{'bf16' if bf16 else 'fp16'} H={H} {geom} cout={co}")
    if fused and co % 4 == 0:
        assert not torch.isnan(r[1]).any(), "a GroupNorm-sum slot was left unwritten"


# ---------------------------------------------------------------------------------------------------------------------------
# LARGE launches of the wave-specialised kernels: thousands of tiles, tile counts that are odd or ragged in x / y, batches whose
# images sit at scales 1e4 apart (an un-normalised launch takes a per-image exponent and a per-image buffer resource), odd chunk
# counts, the 128 + 64-channel tail block.  (Written for round 4's tile-sequence experiment -- a workgroup walking 2 or 4 tiles,
# branch exp-tile-sequences, measured slower and not shipped -- and kept: nothing else in the suite launches more than ~1000
# workgroups of these kernels against a float64 statement.)
SEQ_CASES = [   # (B, H, W, cin, cout, fused)
    (9, 248, 272, 32, 128, False),     # 4743 tiles (odd), 1 chunk
    (9, 248, 272, 96, 128, True),      # ... 3 chunks (odd), GroupNorm + SiLU + temb + residual + sums
    (17, 256, 250, 64, 128, True),     # 8704 tiles, ragged columns, 2 chunks
    (5, 250, 256, 64, 256, False),     # 2560 tiles x 2 channel blocks
    (3, 500, 500, 32, 192, True),      # 6048 tiles x (128 + 64-channel tail block: 2 x 2 wave layout), ragged in x and y
]


@pytest.mark.parametrize("case", SEQ_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_large_launches_of_the_wave_specialised_kernels(cdx_mod, case, dtype):
    ops = cdx_mod.ops
    B, H, W, ci, co, fused = case
    torch.manual_seed(7)
    x = rnd(B, ci, H, W, seed=300) * 1.5 + 0.4
    for b in range(B):                       # images at different scales: the un-normalised launch takes a per-image exponent
        x[b] *= 10.0 ** ((b % 5) - 2)
    w = rnd(co, ci, 3, 3, seed=301, scale=1.0 / math.sqrt(ci * 9))
    bias = rnd(co, seed=302)
    gamma, beta = 1 + 0.2 * rnd(ci, seed=303), 0.3 * rnd(ci, seed=304)
    temb, res = rnd(B, co + 2, seed=305), rnd(B, co, H, W, seed=306)
    half = dtype == "fp16"
    rd = (lambda t: t.half().float()) if half else (lambda t: t)
    h = rd(x).double()
    if fused:
        h = F.silu(F.group_norm(h, 32, gamma.double(), beta.double(), eps=1e-5))
        if half:
            h = h.half().double()
    torch.set_num_threads(16)
    want = F.conv2d(h, rd(w).double(), bias.double(), padding=1)
    if fused:
        want = want + temb[:, 1:1 + co].double()[:, :, None, None] + rd(res).double()
    dt = torch.float16 if half else torch.float32
    big, out = guarded((B, H, W, co), dt)
    out.fill_(float("nan"))
    if half:
        pc = ops.PackedConv16(w.numpy(), bias.numpy(), ci)
        xd = nhwc(x).half()
        kw = dict(out=out)
        if fused:
            sc, sh = ops.gn_stats(xd.float().contiguous(), None, gamma.cuda(), beta.cuda(), 32)
            kw.update(gn=(sc, sh), silu=True, temb=temb.cuda(), temb_off=1, residual=nhwc(res).half(), want_stats=True)
        r = ops.conv16(pc, xd, **kw)
        tol = 4e-3 if fused else 1e-3
    else:
        pc = ops.PackedConv(w.numpy(), bias.numpy(), ci)
        xd = nhwc(x)
        kw = dict(out=out)
        if fused:
            kw.update(gn_affine=(gamma.cuda(), beta.cuda(), 32), silu=True, temb=temb.cuda(), temb_off=1, residual=nhwc(res), want_stats=True)
        r = ops.conv(pc, xd, **kw)
        tol = 3e-6
    torch.cuda.synchronize()
    assert (big[out.numel():] == torch.tensor(SENTINEL).to(dt).item()).all(), "wrote past the output tensor"
    got = nchw(out.float())
    for b in range(B):                       # each image against ITS output scale (the scales differ by 1e4 when un-normalised)
        close(got[b], want[b], tol, f"{dtype} image {b} of {case}")
    if fused:
        st = r[1]
        assert not torch.isnan(st).any(), "a GroupNorm-sum slot was left unwritten"
        close(st[..., 0].sum(1).cpu(), got.double().sum((2, 3)), 2e-3 if half else 1e-5, "GroupNorm sums")
    # determinism: the same launch again gives the same bits
    out2 = torch.empty_like(out)
    kw["out"] = out2
    (ops.conv16 if half else ops.conv)(pc, xd, **kw)
    assert torch.equal(out, out2)
