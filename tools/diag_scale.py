#!/usr/bin/env python3
"""Per-layer error of one UNet forward against the float64 oracle, per numerical variant.

The whole-forward error at input scale 1e-4 was 5x the CPU float32 oracle's in round 3.  Three candidates: (i) the split
operands' 22-23 bits, (ii) bias + temb carried in the accumulators (re-rounded at the bias' ulp by every MFMA), (iii) GroupNorm's
float32 x * scale + shift with a float32 shift.  This tool separates them: for each variant (a subprocess, because the tuning
build reads CDX_DIAG once) it runs the forward of tests/test_range_gpu.py's network at each input scale and records, per
convolution layer, max |hip - float64 oracle| / max |oracle| of that layer's output.

  python tools/diag_scale.py                 # all variants -> gpurun_out/diag_scale.jsonl (+ a table on stdout)
  python tools/diag_scale.py --child NAME    # one variant in this process (env set by the parent)

Variants (libcdx_tune.so; CDX_DIAG bits: 2048 = bias + temb through the accumulator init, 4096 = float32 GroupNorm shift):
  r03_split      split tiles, round 3's numerics (2048 | 4096)
  bias_epilogue  split tiles, bias + temb in the epilogue FMA, float32 shift (4096)
  gn_shift64     split tiles, bias in the accumulators, shift evaluated in float64 (2048)
  r04_split      split tiles, both fixes (0)  = the shipped library
  r03_f32mfma    UNet(split=False): f32-input MFMA kernels, float32 shift (4096)
  r04_f32mfma    UNet(split=False), float64-evaluated shift (0)
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {"r03_split": (True, 2048 | 4096), "bias_epilogue": (True, 4096), "gn_shift64": (True, 2048), "r04_split": (True, 0),
            "r03_f32mfma": (False, 4096), "r04_f32mfma": (False, 0)}
SCALES = (1e-4, 1.0, 1e4)


def rnd(*shape, seed, scale=1.0):
    import torch
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def child(name):
    import torch
    import cdx
    import oracle
    split, _ = VARIANTS[name]
    cfg = cdx.unet_config(image_size=32, base_channels=32, channel_mult=(1, 2, 2), attn_resolutions=(16,), num_res_blocks=1)
    params = cdx.init_params(cfg, seed=5, affine_jitter=0.1)
    net = cdx.UNet(cfg, params, device="cuda:0", split=split)
    t = torch.tensor([500, 17])
    rows = []
    for scale in SCALES:
        x, cond = rnd(2, 3, 32, 32, seed=91, scale=scale), rnd(2, 3, 2, 2, seed=92, scale=scale)
        t64, t32 = {}, {}
        want = oracle.unet_forward_ref(cfg, params, x.double(), t, cond.double(), dtype=torch.float64, taps=t64)
        w32 = oracle.unet_forward_ref(cfg, params, x, t, cond, taps=t32)
        got = net(x.cuda(), t.cuda(), cond.cuda()).cpu()
        layers = {}
        for lname, buf in net.plan(2).outs.items():
            ref = t64[lname]
            g = buf[..., :ref.shape[1]].permute(0, 3, 1, 2).double().cpu()
            den = ref.abs().max().item()
            layers[lname] = {"hip": (g - ref).abs().max().item() / den, "cpu_f32": (t32[lname].double() - ref).abs().max().item() / den}
        den = want.abs().max().item()
        rows.append({"variant": name, "split": split, "cdx_diag": int(os.environ.get("CDX_DIAG", "0")), "scale": scale,
                     "forward_rel_err": (got.double() - want).abs().max().item() / den,
                     "cpu_f32_forward_rel_err": (w32.double() - want).abs().max().item() / den, "layers": layers})
    for r in rows:
        print("DIAG " + json.dumps(r), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "diag_scale.jsonl"))
    args = ap.parse_args()
    if args.child:
        return child(args.child)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    rows = []
    for name, (_, diag) in VARIANTS.items():
        env = dict(os.environ, CDX_TUNE="1", CDX_DIAG=str(diag))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name], env=env, capture_output=True, text=True, timeout=900)
        if r.returncode:
            print(r.stdout[-2000:], r.stderr[-4000:])
            raise SystemExit(f"variant {name} failed")
        rows += [json.loads(ln[5:]) for ln in r.stdout.splitlines() if ln.startswith("DIAG ")]
    with open(args.out, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
    for scale in SCALES:
        sel = [r for r in rows if r["scale"] == scale]
        print(f"\n== input scale {scale:g}: whole forward rel err (CPU float32 oracle: {sel[0]['cpu_f32_forward_rel_err']:.3e})")
        for r in sel:
            print(f"  {r['variant']:14s} {r['forward_rel_err']:.3e}")
        names = list(sel[0]["layers"])
        print("  per layer (first 12 + last 2):  " + "  ".join(f"{r['variant'][:10]:>10s}" for r in sel) + "     cpu_f32")
        for ln in names[:12] + names[-2:]:
            print(f"  {ln:22s} " + "  ".join(f"{r['layers'][ln]['hip']:10.2e}" for r in sel) + f"  {sel[0]['layers'][ln]['cpu_f32']:10.2e}")


if __name__ == "__main__":
    main()
