#!/usr/bin/env python3
"""Time one convolution shape in isolation (HIP events), optionally forcing tile shapes: the A/B harness for
kernel tuning (interleaved rounds in one process, cdna guide rule 24) and the target of rocprofv3 --pmc runs.

  python tools/conv_bench.py [--shape B,H,W,C0,C1,COUT,K,S] [--tiles -1,0,6] [--gn] [--rounds 5] [--iters 10]
"""
import argparse, ctypes, os, sys, math
if any(int(t) >= 16 for a in sys.argv[1:] if a.startswith("--tiles") for t in a.split("=")[-1].split(",") if t.lstrip("-").isdigit()) or \
        any(sys.argv[i] == "--tiles" and any(int(t) >= 16 for t in sys.argv[i + 1].split(",")) for i in range(1, len(sys.argv) - 1)):
    os.environ["CDX_TUNE"] = "1"      # ablation tiles live in libcdx_tune.so (make -C .../csrc EXPERIMENTS=1)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cdx
from cdx import ops, _abi

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="16,256,256,128,0,128,3,1")
ap.add_argument("--tiles", default="-1")
ap.add_argument("--gn", action="store_true", help="fuse GroupNorm scale/shift + SiLU on load + temb + residual (ResBlock conv)")
ap.add_argument("--check", action="store_true", help="compare every tile's output with the first tile's")
ap.add_argument("--stats", action="store_true", help="also produce the GroupNorm partial sums of the output (as every normed layer of the UNet does)")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
B, H, W, c0, c1, co, k, s = map(int, a.shape.split(","))
g = torch.Generator(device="cuda").manual_seed(0)
x0 = torch.randn(B, H, W, c0, device="cuda", generator=g)
x1 = torch.randn(B, H, W, c1, device="cuda", generator=g) if c1 else None
w = (np.random.default_rng(0).standard_normal((co, c0 + c1, k, k)) / math.sqrt((c0 + c1) * k * k)).astype(np.float32)
pc = ops.PackedConv(w, np.zeros(co, np.float32), c0, c1)
ho, wo = (H, W) if s == 1 else ((H + 1) // 2, (W + 1) // 2)
out = torch.empty(B, ho, wo, co, device="cuda")
kw = {}
if a.gn:
    gamma, beta = torch.ones(c0 + c1, device="cuda"), torch.zeros(c0 + c1, device="cuda")
    kw = dict(gn=ops.gn_stats(x0, x1, gamma, beta, 32), silu=True, temb=torch.randn(B, co, device="cuda"),
              residual=torch.randn(B, ho, wo, co, device="cuda"))
args = ops.conv_args(pc, x0, x1, out, stride=s, **kw)
if a.stats:
    _stats_keep = ops.conv_stats_buffer(args, "cuda")
    _stats_big = torch.zeros(4 * _stats_keep.numel(), dtype=torch.float64, device="cuda")      # variants with more slots per tile
    args.stats_out = _stats_big.data_ptr()
flops = 2.0 * B * ho * wo * co * (c0 + c1) * k * k
tiles = [int(t) for t in a.tiles.split(",")]
L = _abi.lib()
st = torch.cuda.current_stream().cuda_stream
tiles = [t for t in tiles if L.cdx_conv_f32_tile(ctypes.byref(args), t, None, 0, st) == 0 or print(f"tile {t}: not built for this shape")]
torch.cuda.synchronize()
if a.check:
    ref = None
    for t in tiles:
        out.zero_()
        assert L.cdx_conv_f32_tile(ctypes.byref(args), t, None, 0, st) == 0
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        else:
            print(f"tile {t}: max |diff| vs tile {tiles[0]} = {(out - ref).abs().max().item():.3e} (scale {ref.abs().max().item():.3f})")
res = {t: [] for t in tiles}
for r in range(a.rounds + 1):
    for t in tiles:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            rc = L.cdx_conv_f32_tile(ctypes.byref(args), t, None, 0, st)
            assert rc == 0, rc
        e1.record()
        torch.cuda.synchronize()
        if r:
            res[t].append(e0.elapsed_time(e1) / a.iters)
for t in tiles:
    ms = sorted(res[t])
    med = ms[len(ms) // 2]
    name = _abi.TILE_NAMES.get(t if t >= 0 else L.cdx_conv_select_tile(ctypes.byref(args)), "?")
    print(f"shape {a.shape} gn={a.gn} tile {t:2d} ({name}): median {med:.4f} ms = {flops/med/1e9:.1f} TF   min {ms[0]:.4f} ms = {flops/ms[0]/1e9:.1f} TF")
