#!/usr/bin/env python3
"""Randomised parity sweep of cdx_conv_f16 (fp16 storage, fp16 MFMA, fp32 accumulate) against float64 torch on the CPU
computed from the SAME fp16-rounded operands: random shapes (ragged, concat, upsample, stride 2, 1x1), random fusion
flags, fp16 or fp32 output.  usage: tools/fuzz_conv16.py [cases] [seed]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import cdx
from cdx import ops

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
nchw = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu()
bad = 0
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
    c0 = int(rng.choice([32, 64, 96, 128])) if concat else int(rng.choice([8, 16, 32, 40, 64, 72, 128, 160]))
    c1 = int(rng.choice([32, 64])) if concat else 0
    co = int(rng.choice([4, 8, 32, 48, 64, 96, 128, 160, 200, 256]))
    ci = c0 + c1
    groups = 8 if ci % 32 else 32
    gn = bool(rng.integers(0, 2)) and ci % groups == 0
    silu = gn and bool(rng.integers(0, 2))
    use_temb, use_res = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    out32 = bool(rng.integers(0, 4) == 0)
    want_stats = bool(rng.integers(0, 2)) and not out32
    g = torch.Generator().manual_seed(5000 + case)
    x0 = (torch.randn(B, c0, H, W, generator=g) * 1.5 + 0.3).half()
    x1 = (torch.randn(B, c1, H, W, generator=g) - 0.5).half() if c1 else None
    w = (torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k)).half()
    bias = torch.randn(co, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(ci, generator=g), 0.3 * torch.randn(ci, generator=g)
    xc = (torch.cat([x0, x1], 1) if c1 else x0).double()
    s0, s1 = nhwc(x0), (nhwc(x1) if c1 else None)
    kw = dict(stride=s, upsample=up)
    h = xc
    if gn:
        sc, sh = ops.gn_stats(s0.float(), None if s1 is None else s1.float(), gamma.cuda(), beta.cuda(), groups)
        kw["gn"] = (sc, sh)
        kw["silu"] = silu
        # the kernel applies x*scale + shift in fp32, then SiLU, then rounds the staged value to fp16
        h = xc * nchw(sc[:, None, None, :].expand(B, 1, 1, ci)).double() + nchw(sh[:, None, None, :].expand(B, 1, 1, ci)).double()
        h = F.silu(h) if silu else h
    h = h.float().half().double()                       # the staged LDS image is fp16
    h = F.interpolate(h, scale_factor=2, mode="nearest") if up else h
    want = F.conv2d(h, w.double(), bias.double(), stride=s, padding=k // 2)
    ho, wo = want.shape[-2:]
    temb = torch.randn(B, co + 3, generator=g) if use_temb else None
    res = torch.randn(B, co, ho, wo, generator=g).half() if use_res and not out32 else None
    if use_temb: want = want + temb.double()[:, 1:1 + co, None, None]
    if res is not None: want = want + res.double()
    if use_temb: kw.update(temb=temb.cuda(), temb_off=1)
    if res is not None: kw["residual"] = nhwc(res)
    pc = ops.PackedConv16(w.float().numpy(), bias.numpy(), c0, c1)
    if _trace:
        _trace.write(f"case {case} " + repr(dict(B=B, c0=c0, c1=c1, co=co, H=H, W=W, k=k, s=s, up=up, gn=gn, silu=silu, temb=use_temb, res=res is not None, out32=out32, stats=want_stats)) + "\n"); _trace.flush(); torch.cuda.synchronize()
    try:
        r = ops.conv16(pc, s0, s1, out_dtype=torch.float32 if out32 else torch.float16, want_stats=want_stats, **kw)
        out = r[0] if want_stats else r
    except Exception as e:
        print("case", case, "EXC", repr(e)[:140], dict(B=B, c0=c0, c1=c1, co=co, H=H, W=W, k=k, s=s, up=up, gn=gn, out32=out32, stats=want_stats)); bad += 1; continue
    got = nchw(out.float()).double()
    err = (got - want).abs().max().item() if not torch.isnan(got).any() else float("inf")
    ref = max(want.abs().max().item(), 1.0)
    tol = (2e-5 if out32 else 1.2e-3) * ref + (2e-3 * ref if gn else 0)      # GN path: the staged value's fp16 rounding can flip with fp32 vs fp64 arithmetic
    if err > tol:
        bad += 1
        print("case", case, "BAD", err, "ref", ref, dict(B=B, c0=c0, c1=c1, co=co, H=H, W=W, k=k, s=s, up=up, gn=gn, silu=silu, temb=use_temb, res=res is not None, out32=out32, stats=want_stats))
print("fuzz_conv16:", bad, "bad of", ncases)
