#!/usr/bin/env python3
"""Where do the waves of the wave-specialised split tile wait?  ("32 % of wave cycles parked at the one
barrier per chunk" is a counter over ALL eight waves; this separates the MFMA waves' share from the producers'.)

Runs the SHIPPED 8 x 16-pixel tile with s_memtime stamps (libcdx_tune.so, tile id 108 = variant 48; 109: producers stage only the
first chunk) on one fused ResBlock shape and prints, per role, the share of a wave's life spent inside the per-chunk
__syncthreads, plus the MFMA waves' per-chunk durations.

  python tools/ws_stamps.py [--shape 16,256,256,128,0,128] [--tile 108]
"""
import argparse, ctypes, math, os, sys
os.environ["CDX_TUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cdx
from cdx import ops, _abi

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="16,256,256,128,0,128")
ap.add_argument("--tile", type=int, default=108)
ap.add_argument("--abl", type=int, default=10, help="--fp16: tuning flag value (10 stamps; 11 .. 15 stamped ablations: no staging / LDS reads / weight refills / epilogue / the first three)")
ap.add_argument("--nores", action="store_true", help="no residual (a ResBlock's conv1: GroupNorm + SiLU + temb only)")
ap.add_argument("--fp16", action="store_true", help="the 16-bit storage tile (cdx_conv_f16, tuning flag bits 8..11 = 10) instead of the float32 split tile")
a = ap.parse_args()
B, H, W, c0, c1, co = map(int, a.shape.split(","))
g = torch.Generator(device="cuda").manual_seed(0)
L = _abi.lib()
st = torch.cuda.current_stream().cuda_stream
nblocks = B * ((H + 7) // 8) * ((W + 15) // 16) * ((co + 127) // 128)
rows = 2 * nblocks * 4
big = torch.zeros(max(rows * 16, 1 << 20), dtype=torch.float64, device="cuda")
w = (np.random.default_rng(0).standard_normal((co, c0 + c1, 3, 3)) / math.sqrt((c0 + c1) * 9)).astype(np.float32)
if a.fp16:
    assert c1 == 0
    x0 = torch.randn(B, H, W, c0, device="cuda", generator=g).half()
    pc = ops.PackedConv16(w, np.zeros(co, np.float32), c0, 0)
    out = torch.empty(B, H, W, co, device="cuda", dtype=torch.float16)
    sc, sh = torch.ones(B, c0, device="cuda"), torch.zeros(B, c0, device="cuda")
    args = ops.conv16_args(pc, x0, None, out, gn=(sc, sh), silu=True, temb=torch.randn(B, co, device="cuda"),
                           residual=None if a.nores else torch.randn(B, H, W, co, device="cuda").half())
    args.flags |= a.abl << 8                         # tuning flag bits 8..11: the stamping variant of the 16-bit tile (or a stamped ablation)
    _keep = ops.conv16_stats_buffer(args, "cuda")
    args.stats_out = big.data_ptr()
    launch = lambda: L.cdx_conv_f16(ctypes.byref(args), None, 0, st)      # noqa: E731
else:
    x0 = torch.randn(B, H, W, c0, device="cuda", generator=g)
    x1 = torch.randn(B, H, W, c1, device="cuda", generator=g) if c1 else None
    pc = ops.PackedConv(w, np.zeros(co, np.float32), c0, c1)
    out = torch.empty(B, H, W, co, device="cuda")
    gamma, beta = torch.ones(c0 + c1, device="cuda"), torch.zeros(c0 + c1, device="cuda")
    kw = dict(gn=ops.gn_stats(x0, x1, gamma, beta, 32, act_exp="auto"), silu=True, temb=torch.randn(B, co, device="cuda"),
              residual=None if a.nores else torch.randn(B, H, W, co, device="cuda"))
    args = ops.conv_args(pc, x0, x1, out, **kw)
    _keep = ops.conv_stats_buffer(args, "cuda")      # sets stats_slots
    args.stats_out = big.data_ptr()
    launch = lambda: L.cdx_conv_f32_tile(ctypes.byref(args), a.tile, None, 0, st)      # noqa: E731
for _ in range(3):
    big.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = launch()
    assert rc == 0, rc
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
T = big.view(torch.int64)[: rows * 16].reshape(rows, 16).cpu().numpy().astype(np.int64)
M, P = T[: nblocks * 4], T[nblocks * 4:]
nch = (c0 + c1 + 31) // 32
life = M[:, 12] - M[:, 0]
# tick length: s_memtime counters of different XCDs / SEs have different bases, so the launch span cannot be read off them; with two
# workgroups resident on each of the 256 CUs for the whole launch, mean life x workgroups = 512 x launch time
tick = ms * 1e6 * 512 / (nblocks * life.mean())
rt = (M[:, 10] - M[:, 11]).astype(np.float64)          # 100 MHz ticks over the same span
okc = rt > 0
clk = np.median(life[okc] / rt[okc]) * 0.1            # GHz
tick = 1.0 / clk
print(f"tile {a.tile}: launch {ms:.4f} ms, {nblocks} workgroups, {nch} chunks; in-kernel clock (s_memtime / s_memrealtime, median over waves) {clk:.3f} GHz "
      f"(p10 {np.percentile(life[okc] / rt[okc], 10) * 0.1:.3f}, p90 {np.percentile(life[okc] / rt[okc], 90) * 0.1:.3f}); "
      f"mean residency implied by 2 workgroups per CU: {ms * 1e3 * 512 / nblocks:.1f} us per workgroup")
print(f"MFMA waves: life {life.mean() * tick / 1e3:.2f} us; inside the per-chunk barriers {M[:, 15].mean() * tick / 1e3:.2f} us = {M[:, 15].sum() / life.sum():.3f} of their life "
      f"(p10 {np.percentile(M[:, 15] / life, 10):.3f}, p90 {np.percentile(M[:, 15] / life, 90):.3f})")
seg = {"entry -> first chunk staged (barrier passed)": M[:, 2] - M[:, 0]}
for c in range(min(nch, 8)):
    seg[f"chunk {c} (MFMAs + barrier)"] = M[:, 3 + c] - M[:, 2 + c]
seg["epilogue (stores issued)"] = M[:, 12] - M[:, 2 + min(nch, 8)]
for k, v in seg.items():
    print(f"  {k:46s} mean {v.mean() * tick / 1e3:7.3f} us   share {v.sum() / life.sum():.3f}")
plife = P[:, 3] - P[:, 0]
ok = plife > 0
if (P[ok, 4] > 0).all():      # (builds that stamp the producers' prologue)
    for name, a_, b_ in (("entry -> first halo loads issued", 0, 4), ("... -> first pass staged (its loads are back)", 4, 5), ("... -> first chunk staged", 5, 6), ("... -> first barrier passed", 6, 7)):
        v = (P[ok, b_] - P[ok, a_])
        print(f"  producers: {name:48s} mean {v.mean() * tick / 1e3:7.3f} us")
print(f"producer waves: life {plife[ok].mean() * tick / 1e3:.2f} us; at barriers {P[ok, 1].sum() / plife[ok].sum():.3f}, staging (LDS writes + next loads issued) {P[ok, 2].sum() / plife[ok].sum():.3f}")
