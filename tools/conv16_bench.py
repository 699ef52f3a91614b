#!/usr/bin/env python3
"""Time the fp16-storage convolution on one shape, with the timing-only ablation variants (flag bits 8..10)."""
import argparse, ctypes, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if "--shipped" not in sys.argv:
    os.environ["CDX_TUNE"] = "1"      # ablation variants live in libcdx_tune.so (make EXPERIMENTS=1)
import cdx
from cdx import ops, _abi
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="16,256,256,128,0,128,3,1")
ap.add_argument("--abl", default="0,1,2,3,4,7")
ap.add_argument("--plain", action="store_true")
ap.add_argument("--shipped", action="store_true", help="time libcdx.so (abl 0 only) instead of the tuning build")
ap.add_argument("--nores", action="store_true")
ap.add_argument("--nostats", action="store_true")
a = ap.parse_args()
B, H, W, c0, c1, co, k, s = map(int, a.shape.split(","))
x0 = torch.randn(B, H, W, c0, device="cuda").half()
x1 = torch.randn(B, H, W, c1, device="cuda").half() if c1 else None
w = (np.random.default_rng(0).standard_normal((co, c0 + c1, k, k)) / math.sqrt((c0 + c1) * k * k)).astype(np.float32)
pc = ops.PackedConv16(w, np.zeros(co, np.float32), c0, c1)
out = torch.empty(B, H // s, W // s, co, device="cuda", dtype=torch.float16)
kw = {}
if not a.plain:
    sc, sh = torch.ones(B, c0 + c1, device="cuda"), torch.zeros(B, c0 + c1, device="cuda")
    kw = dict(gn=(sc, sh), silu=True, temb=torch.randn(B, co, device="cuda"))
    if not a.nores:
        kw["residual"] = torch.randn(B, H // s, W // s, co, device="cuda").half()
args = ops.conv16_args(pc, x0, x1, out, stride=s, **kw)
stats = ops.conv16_stats_buffer(args, "cuda") if not (a.plain or a.nostats) else None
flops = 2.0 * B * (H // s) * (W // s) * co * (c0 + c1) * k * k
L = _abi.lib(); st = torch.cuda.current_stream().cuda_stream
base = args.flags
for r in range(3):
    for abl in [int(v) for v in a.abl.split(",")]:
        args.flags = base | (abl << 8)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            rc = L.cdx_conv_f16(ctypes.byref(args), None, 0, st); assert rc == 0, rc
        e1.record(); torch.cuda.synchronize()
        if r == 2:
            ms = e0.elapsed_time(e1) / 10
            print(f"conv16 {a.shape} plain={a.plain} nores={a.nores} nostats={a.nostats} abl={abl} (1 no-epilogue, 2 no-staging, 4 no-weight-refill): {ms:.4f} ms = {flops/ms/1e9:.0f} TF")
