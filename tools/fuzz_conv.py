#!/usr/bin/env python3
"""Randomised parity sweep of cdx_conv_f32 (library's own tile choice) against float64 torch on the CPU:
random layer shapes (incl. ragged sizes, concat, upsample, stride 2, 1x1, tiny / huge channel counts), random fusion
flags (GroupNorm+SiLU on load, temb, residual, GroupNorm sums of the output) and -- VERDICT r02 item 1 -- a random
log-uniform SCALE per source (10^U(-6, 6)) and, independently, per additive term (bias, temb, residual: 10^U(-3, 3) times
the products' scale): the error is judged relative to the output's own scale, no floor.
usage: tools/fuzz_conv.py [cases] [seed] [convout]      (convout: only layers the cout <= 3 GEMM form, tile 12, can take)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import cdx
from cdx import ops, _abi

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
convout = len(sys.argv) > 3 and sys.argv[3] == "convout"
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
nchw = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu()
bad, tiles_seen = 0, {}
_trace = open(os.environ['FUZZ_TRACE'], 'a') if os.environ.get('FUZZ_TRACE') else None      # per-case log, flushed BEFORE the launch
for case in range(ncases):
    k = int(rng.choice([3, 3, 3, 1]))
    s = int(rng.choice([1, 1, 1, 2])) if k == 3 else 1
    up = bool(rng.integers(0, 4) == 0) and s == 1 and k == 3
    B = int(rng.integers(1, 4))
    H, W = int(rng.integers(2, 41)), int(rng.integers(2, 73))
    if up:
        H, W = max(1, H // 2), max(1, W // 2)
    concat = bool(rng.integers(0, 3) == 0)
    c0 = int(rng.choice([32, 64, 96, 128])) if concat else int(rng.choice([4, 8, 12, 32, 40, 64, 72, 128, 160]))
    c1 = int(rng.choice([32, 64])) if concat else 0
    co = int(rng.choice([1, 2, 3, 4, 8, 32, 48, 64, 96, 128, 160, 200, 256]))
    if convout:      # 3x3 stride 1, ONE source of 64 / 128 / 192 / 256 channels, cout <= 3, at least 32 pixels wide, any height
        k, s, up, concat, c1 = 3, 1, False, False, 0
        c0, co = int(rng.choice([64, 128, 192, 256])), int(rng.integers(1, 4))
        H, W = int(rng.integers(1, 41)), int(rng.integers(32, 101))
    ci = c0 + c1
    groups = 4 if ci % 32 else 32
    gn = bool(rng.integers(0, 2)) and ci % groups == 0
    silu = gn and bool(rng.integers(0, 2))
    use_temb, use_res = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    want_stats = bool(rng.integers(0, 2)) and co % 4 == 0
    g = torch.Generator().manual_seed(1000 + case)
    sx0, sx1 = 10.0 ** rng.uniform(-6, 6), 10.0 ** rng.uniform(-6, 6)
    x0 = (torch.randn(B, c0, H, W, generator=g, dtype=torch.float64) * 1.5 + 0.3) * sx0
    x1 = (torch.randn(B, c1, H, W, generator=g, dtype=torch.float64) - 0.5) * sx1 if c1 else None
    w = torch.randn(co, ci, k, k, generator=g, dtype=torch.float64) / math.sqrt(ci * k * k)
    bias = torch.randn(co, generator=g, dtype=torch.float64)
    gamma, beta = 1 + 0.2 * torch.randn(ci, generator=g, dtype=torch.float64), 0.3 * torch.randn(ci, generator=g, dtype=torch.float64)
    xc = torch.cat([x0, x1], 1) if c1 else x0
    xf = xc.float().double()                         # the kernel sees float32 inputs
    h = F.group_norm(xf, groups, gamma.float().double(), beta.float().double(), eps=1e-5) if gn else xf
    h = F.silu(h) if silu else h
    h = F.interpolate(h, scale_factor=2, mode="nearest") if up else h
    want = F.conv2d(h, w.float().double(), bias.float().double(), stride=s, padding=k // 2)
    ho, wo = want.shape[-2:]
    # additive terms: each at its OWN scale, 10^U(-3, 3) times the scale of the products (drawn "at the scale of
    # the products" the sweep could not see a bias that dwarfs them -- conv_in's 0.1 over a 1e-4 signal)
    psc = 1.0 if gn else max(sx0, sx1 if c1 else 0.0)
    osc_b, osc_t, osc_r = (psc * 10.0 ** rng.uniform(-3, 3) for _ in range(3))
    want = want + (bias.float().double() * (osc_b - 1.0))[None, :, None, None]
    bias = bias * osc_b
    temb = torch.randn(B, co + 3, generator=g, dtype=torch.float64) * osc_t if use_temb else None
    res = torch.randn(B, co, ho, wo, generator=g, dtype=torch.float64) * osc_r if use_res else None
    if use_temb: want = want + temb.float().double()[:, 1:1 + co, None, None]
    if use_res: want = want + res.float().double()
    s0, s1 = nhwc(x0.float()), (nhwc(x1.float()) if c1 else None)
    kw = dict(stride=s, upsample=up)
    if gn:
        kw["gn_affine"] = (gamma.float().cuda(), beta.float().cuda(), groups)      # GroupNorm with the exponent the chosen tile wants
        kw["silu"] = silu
    if use_temb: kw.update(temb=temb.float().cuda(), temb_off=1)
    if use_res: kw["residual"] = nhwc(res.float())
    pc = ops.PackedConv(w.float().numpy(), bias.float().numpy(), c0, c1, split=(case % 3 != 0))      # every third case: f32-MFMA kernels only
    if _trace:
        _trace.write(f"case {case} " + repr(dict(B=B, c0=c0, c1=c1, co=co, H=H, W=W, k=k, s=s, up=up, gn=gn, silu=silu, temb=use_temb, res=use_res, stats=want_stats, split=(case % 3 != 0))) + "\n"); _trace.flush(); torch.cuda.synchronize()
    try:
        out = torch.full((B, ho, wo, co), float("nan"), device="cuda")
        if want_stats:
            _, st = ops.conv(pc, s0, s1, out=out, want_stats=True, **kw)
        else:
            ops.conv(pc, s0, s1, out=out, **kw)
        kq = {k: v for k, v in kw.items() if k != "gn_affine"}
        if gn:
            kq["gn"] = (torch.empty(B, ci, device="cuda"), torch.empty(B, ci, device="cuda"), 1)
        elif case % 3 != 0:
            kq["src_amax"] = (ops.amax_buffer(B, "cuda"),) * (2 if c1 else 1)
        tile = _abi.lib().cdx_conv_select_tile(__import__("ctypes").byref(ops.conv_args(pc, s0, s1, out, **kq)))
        if tile < 0 and gn:      # (a GroupNorm exponent on a launch that does not take the split tile is refused: ask without it)
            kq["gn"] = kq["gn"][:2]
            tile = _abi.lib().cdx_conv_select_tile(__import__("ctypes").byref(ops.conv_args(pc, s0, s1, out, **kq)))
    except Exception as e:
        print("case", case, "EXC", repr(e)[:120], dict(B=B, c0=c0, c1=c1, co=co, H=H, W=W, k=k, s=s, up=up, gn=gn, stats=want_stats)); bad += 1; continue
    tiles_seen[tile] = tiles_seen.get(tile, 0) + 1
    got = nchw(out).double()
    err = (got - want).abs().max().item() if not torch.isnan(got).any() else float("inf")
    ref = max(want.abs().max().item(), 1e-300)
    ok = err <= 6e-6 * ref
    if ok and want_stats:
        # the sums left by the epilogue reproduce the moments of the stored tensor
        hw = ho * wo
        sc, sh, m, r = ops.gn_finalize(st, None, hw, torch.ones(co, device="cuda"), torch.zeros(co, device="cuda"), 4 if co % 32 else 32, want_moments=True)
        gg = 4 if co % 32 else 32
        xg = got.reshape(B, gg, -1)
        e1 = (m.cpu().double() - xg.mean(-1)).abs().max().item()
        ok = e1 <= 2e-6 * ref
        if not ok: err = ("stats", e1)
    if not ok:
        bad += 1
        print("case", case, "BAD", err, "tile", _abi.TILE_NAMES.get(tile, tile), dict(B=B, c0=c0, c1=c1, co=co, H=H, W=W, k=k, s=s, up=up, gn=gn, silu=silu, temb=use_temb, res=use_res, stats=want_stats))
print("fuzz_conv:", bad, "bad of", ncases, "tiles used:", {(_abi.TILE_NAMES.get(t, t)): n for t, n in sorted(tiles_seen.items())})
