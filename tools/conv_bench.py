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
ap.add_argument("--stamps", type=int, default=0, help="tile id of a stamping variant (92): run it once and digest the per-wave phase stamps")
ap.add_argument("--check", action="store_true", help="compare every tile's output with the first tile's")
ap.add_argument("--stats", action="store_true", help="also produce the GroupNorm partial sums of the output (as every normed layer of the UNet does)")
ap.add_argument("--no-amax", action="store_true", help="with --stats: do not request amax_out")
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
_tiles = [int(t) for t in a.tiles.split(",")]
if a.gn:
    gamma, beta = torch.ones(c0 + c1, device="cuda"), torch.zeros(c0 + c1, device="cuda")
    # split tiles only: GroupNorm scale / shift pre-multiplied by the static activation exponent, as the UNet plan does
    exp = "auto" if all(t == 11 or t >= 60 for t in _tiles) else None
    kw = dict(gn=ops.gn_stats(x0, x1, gamma, beta, 32, act_exp=exp), silu=True, temb=torch.randn(B, co, device="cuda"),
              residual=torch.randn(B, ho, wo, co, device="cuda"))
else:      # un-normalised launch: the split tile takes its exponent from the per-image maxima of the sources
    kw = dict(src_amax=(ops.amax(x0),) + ((ops.amax(x1),) if x1 is not None else ()))
args = ops.conv_args(pc, x0, x1, out, stride=s, **kw)
if a.stats and not a.no_amax:      # ... and leaves the maxima of its output, as every block output of the UNet does
    _amax_keep = ops.amax_buffer(B, "cuda")
    args.amax_out = _amax_keep.data_ptr()
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
if a.stamps:
    assert a.stats, "--stamps needs --stats (the stamps go to the stats buffer)"
    for _ in range(3):
        _stats_big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert L.cdx_conv_f32_tile(ctypes.byref(args), a.stamps, None, 0, st) == 0
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    nw = B * ((ho + 3) // 4) * ((wo + 31) // 32) * ((co + 127) // 128) * 4
    T = _stats_big.view(torch.int64)[: nw * 16].reshape(nw, 16).cpu().numpy().astype(np.int64)
    np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"stamps{a.stamps}.npy"), T)
    xcc = T[:, 14] & 0xF
    span = max(T[xcc == x][:, 12].max() - T[xcc == x][:, 0].min() for x in np.unique(xcc))     # (each XCD has its own counter base)
    tick_ns = ms * 1e6 / span
    nch = (c0 + c1 + 31) // 32
    print(f"stamps: {nw} waves, launch {ms:.4f} ms (event), span {span} ticks -> {tick_ns:.2f} ns per tick")
    life = (T[:, 12] - T[:, 0]) * tick_ns / 1e3
    seg = {"entry->loads issued": T[:, 1] - T[:, 0], "loads issued->chunk0 staged+barrier": T[:, 2] - T[:, 1]}
    for c in range(min(nch, 8)):
        seg[f"chunk {c}"] = T[:, 3 + c] - T[:, 2 + c]
    seg["epilogue (stores issued)"] = T[:, 12] - T[:, 2 + min(nch, 8)]
    print(f"wave lifetime us: mean {life.mean():.2f}  p10 {np.percentile(life, 10):.2f}  p50 {np.percentile(life, 50):.2f}  p90 {np.percentile(life, 90):.2f}")
    for k, v in seg.items():
        v = v * tick_ns / 1e3
        print(f"  {k:38s} mean {v.mean():7.3f} us  p10 {np.percentile(v, 10):7.3f}  p50 {np.percentile(v, 50):7.3f}  p90 {np.percentile(v, 90):7.3f}   share {v.mean() / life.mean():.3f}")
    hw = T[:, 13]
    simd_key = hw & 0xFFFFFFF0 & ~(0xF << 0)      # everything but the wave slot id (bits 3:0)
    import collections
    by = collections.defaultdict(list)
    for i in range(nw):
        by[(int(T[i, 14]) & 0xF, (int(hw[i]) >> 4) & 3, (int(hw[i]) >> 8) & 0xFF)].append((int(T[i, 0]), int(T[i, 12]), int(T[i, 2]), int(T[i, 2 + min(nch, 8)])))
    conc, mf2, mf1, mf0, tot = [], 0, 0, 0, 0
    for k, ws in by.items():
        ev = []
        for s0, s1, m0, m1 in ws:
            ev += [(m0, 1), (m1, -1)]
        ev.sort()
        cur, last = 0, ev[0][0]
        for t, d in ev:
            dt = t - last
            if cur >= 2: mf2 += dt
            elif cur == 1: mf1 += dt
            else: mf0 += dt
            cur += d
            last = t
    tot = mf0 + mf1 + mf2
    print(f"per SIMD (XCC, SE/SH/CU, SIMD): {len(by)} groups, {nw / len(by):.1f} waves each; time with >=2 / 1 / 0 waves inside their chunk loops: {mf2 / tot:.3f} / {mf1 / tot:.3f} / {mf0 / tot:.3f}")
    sys.exit(0)
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
