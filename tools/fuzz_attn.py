#!/usr/bin/env python3
"""Randomised parity sweep of cdx_attn_f32 / cdx_attn_f16 and cdx_linear_f32 against float64 torch on the CPU: ragged
query / key counts (self- and cross-attention shapes), 1..12 heads, large-magnitude scores.  usage: tools/fuzz_attn.py [cases] [seed]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cdx
from cdx import ops

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(ncases):
    B, heads = int(rng.integers(1, 4)), int(rng.integers(1, 13))
    nq = int(rng.choice([1, 5, 31, 32, 33, 64, 100, 128, 129, 256, 300, 1024]))
    nk = int(rng.choice([1, 7, 31, 32, 33, 64, 77, 256, 257, 1024]))
    half = bool(rng.integers(0, 3) == 0)
    scale = float(rng.choice([0.3, 1.0, 3.0]))          # large scores: softmax must subtract the row maximum
    c = heads * 64
    g = torch.Generator().manual_seed(9000 + case)
    q, k, v = (torch.randn(B, n, c, generator=g) * scale for n in (nq, nk, nk))
    dt = torch.float16 if half else torch.float32
    qd, kd, vd = q.to(dt), k.to(dt), v.to(dt)
    got = ops.attention(qd.cuda(), kd.cuda(), vd.cuda(), heads).float().cpu().double()
    Q, K, V = (t.double().reshape(B, -1, heads, 64).transpose(1, 2) for t in (qd, kd, vd))
    want = (torch.softmax(Q @ K.transpose(-1, -2) / 8.0, -1) @ V).transpose(1, 2).reshape(B, nq, c)
    err = (got - want).abs().max().item() if not torch.isnan(got).any() else float("inf")
    ref = max(want.abs().max().item(), 1.0)
    if err > (1.5e-3 if half else 5e-6) * ref:
        bad += 1
        print("case", case, "BAD attn", err, dict(B=B, heads=heads, nq=nq, nk=nk, half=half, scale=scale))
    # linear: M <= 64 rows, any N / K
    M, N, K = int(rng.integers(1, 65)), int(rng.choice([3, 64, 100, 512, 2048])), int(rng.choice([4, 128, 130, 512]))
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    si = bool(rng.integers(0, 2))
    try:
        got = ops.linear(x.cuda(), w.cuda(), b.cuda(), silu_in=si).cpu().double()
        xi = torch.nn.functional.silu(x.double()) if si else x.double()
        want = xi @ w.double().T + b.double()
        err = (got - want).abs().max().item()
        if err > 5e-6 * max(want.abs().max().item(), 1.0):
            bad += 1
            print("case", case, "BAD linear", err, dict(M=M, N=N, K=K, silu=si))
    except cdx._abi.CdxError as e:
        if "EINVAL" not in str(e) and "ENOTSUP" not in str(e):
            bad += 1
            print("case", case, "EXC linear", e)
print("fuzz_attn:", bad, "bad of", ncases)
