#!/usr/bin/env python3
"""bench.py -- decoded images/s of the reverse-diffusion hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: one UNet forward (eps prediction) + one DDIM
update for the BASELINE.json configs[1] workload -- 256x256x3, 128-ch UNet with self-attention at 16^2,
batch 16 per GPU, float32.  `value` = images decoded per second by the whole job =
(16 * N images) / (100 steps * seconds-per-step): the metric is quoted on 100-step DDIM, every step has
identical cost, and K defaults to 100 (one complete decode).  Inputs (cond, x_T) are resident in HBM
before the timed region.  One process per GPU; images are independent, so there is no data-path
collective (torch.distributed is used only for the barrier and the max-over-ranks of the elapsed time).

Extra objects on the JSON line:
  roofline     -- the dominant kernel symbol (the implicit-GEMM conv instantiation carrying most FLOPs):
                  algorithmic FLOPs of its launches in one forward / their summed durations, measured with
                  HIP events on the launch stream, against the fp32 MFMA peak (157.3 TFLOP/s).
  cpu_baseline -- the stock-torch CPU oracle (oracle/, kind "port": the reference ships no sampler) timed
                  on this host's cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (matrix)" (spec; 155 measured)
FP16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"


def conv_variant(a) -> tuple:
    """(ksize, stride, log2 TW, tile shape) of the kernel instantiation libcdx picks for this launch."""
    import ctypes
    import cdx
    logtw = 5 if a.wout >= 32 else 4 if a.wout >= 16 else 3 if a.wout >= 8 else 2
    if isinstance(a, cdx._abi.ConvF16Args):
        return (a.ksize, a.stride, logtw, "f16_64x128" if a.stride == 2 else "f16_128x128")
    tile = cdx._abi.lib().cdx_conv_select_tile(ctypes.byref(a))
    return (a.ksize, a.stride, logtw, cdx._abi.TILE_NAMES[tile])


def conv_flops(a) -> float:
    return 2.0 * a.batch * a.hout * a.wout * a.cout * (a.c0 + a.c1) * a.ksize * a.ksize


def conv_bytes(a, elem=4) -> float:
    """Algorithmic HBM bytes of one conv launch: every input, residual and output element once, weights once."""
    px_in = a.batch * a.hin * a.win
    px_out = a.batch * a.hout * a.wout
    b = px_in * (a.c0 + a.c1) * elem + px_out * a.cout * elem + a.cout * (a.c0 + a.c1) * a.ksize * a.ksize * elem
    if a.residual:
        b += px_out * a.cout * elem
    return float(b)


def plan_flops(plan) -> dict:
    """Algorithmic FLOPs of one forward, per op family (SURVEY.md section 8d formulas)."""
    out = {"conv": 0.0, "attn": 0.0, "linear": 0.0}
    for fn, a, _, _ in plan.calls:
        n = fn.__name__
        if n in ("cdx_conv_f32", "cdx_conv_f16"):
            out["conv"] += conv_flops(a)
        elif n in ("cdx_attn_f32", "cdx_attn_f16"):
            out["attn"] += 4.0 * a.batch * a.heads * a.nq * a.nk * a.head_dim
        elif n == "cdx_linear_f32":
            out["linear"] += 2.0 * a.m * a.n * a.k
    return out


def fn_is_half(a) -> bool:
    return type(a).__name__ == "ConvF16Args"


def measure_dominant_kernel(plan, torch, reps=3):
    """Per-launch HIP-event timing of every conv launch of one forward; returns the roofline object for the
    kernel symbol with the most FLOPs, plus a per-variant table."""
    import ctypes
    st = torch.cuda.current_stream()
    convs = [(i, c) for i, c in enumerate(plan.calls) if c[0].__name__ == "cdx_conv_f32"]
    table, shapes = {}, {}
    for rep in range(reps + 1):
        evs = []
        for i, (fn, a, wp, wb) in enumerate(plan.calls):
            if fn.__name__ in ("cdx_conv_f32", "cdx_conv_f16"):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                fn(ctypes.byref(a), wp, wb, st.cuda_stream)
                e1.record(st)
                evs.append((a, e0, e1))
            else:
                fn(ctypes.byref(a), wp, wb, st.cuda_stream)
        torch.cuda.synchronize()
        if rep == 0:
            continue   # warm-up
        for a, e0, e1 in evs:
            ms = e0.elapsed_time(e1)
            t = table.setdefault(conv_variant(a), {"flops": 0.0, "ms": 0.0, "launches": 0, "bytes": 0.0})
            t["flops"] += conv_flops(a)
            t["bytes"] += conv_bytes(a, 2 if fn_is_half(a) else 4)
            t["ms"] += ms
            t["launches"] += 1
            sk = "k%ds%d %3dx%-3d %4d->%-4d %s" % (a.ksize, a.stride, a.hout, a.wout, a.c0 + a.c1, a.cout, conv_variant(a)[3])
            u = shapes.setdefault(sk, {"flops": 0.0, "ms": 0.0, "n": 0})
            u["flops"] += conv_flops(a); u["ms"] += ms; u["n"] += 1
    dom = max(table, key=lambda k: table[k]["flops"])
    d = table[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    wino = dom[3].startswith("wino")
    half = dom[3].startswith("f16")
    peak = FP16_MFMA_PEAK_TFLOPS if half else FP32_MFMA_PEAK_TFLOPS
    # Winograd F(2x2,3x3) issues 16 multiply-adds where the direct algorithm (which `achieved` counts) needs 36.
    executed = achieved / 2.25 if wino else achieved
    roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None,
            "kernel": ("conv_wino8_kernel<WinoCfg<2,0>> (Winograd F(2x2,3x3), ksize %d stride %d log2TW %d tile %s)" if wino else
                       "conv16_kernel<Conv16Cfg> (fp16 MFMA, ksize %d stride %d log2TW %d tile %s)" if half else
                       "conv_kernel<ConvCfg<ksize %d, stride %d, log2TW %d, tile %s>>") % dom,
            "flop_accounting": "achieved = ALGORITHMIC direct-convolution FLOPs (2*Cin*Cout*9*H*W*B) / measured time"
                               + ("; this kernel is Winograd F(2x2,3x3): it issues 2.25x fewer MFMA FLOPs than that, so "
                                  "frac can exceed 1; executed_* is the MFMA work actually issued" if wino else ""),
            "executed_tflops": round(executed, 2), "executed_frac": round(executed / peak, 4),
            "avg_launch_ms": round(d["ms"] / d["launches"], 4), "launches_per_forward": d["launches"] // reps,
            "flop_share_of_forward": round(d["flops"] / sum(v["flops"] for v in table.values()), 4),
            "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
            "hbm_GBps_at_algorithmic_bytes": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1)}
    # HBM bytes per launch of this kernel: PMC counters cannot be read from inside the process, so the figure
    # comes from the committed rocprofv3 --pmc passes over this same command (profiles/r01_traffic.json).
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if wino and "wino" in tr["kernel"]:
            roof["traffic"] = round(tr["hbm_bytes_per_launch"])
            roof["traffic_over_algorithmic"] = round(tr["hbm_bytes_per_launch"] / roof["algorithmic_bytes_per_launch"], 3)
            roof["traffic_source"] = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 2 x FETCH + WRITE)"
    except (OSError, KeyError, ValueError):
        pass
    per_variant = {"k%ds%d_tw%d_%s" % k: {"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                         "ms_per_forward": round(v["ms"] / reps, 3)}
                   for k, v in sorted(table.items(), key=lambda kv: -kv[1]["flops"])}
    per_shape = {k: {"launches": v["n"] // reps, "ms_per_forward": round(v["ms"] / reps, 3),
                     "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)}
                 for k, v in sorted(shapes.items(), key=lambda kv: -kv[1]["ms"])}
    return roof, {"variants": per_variant, "shapes": per_shape}


def host_cores() -> int:
    """Cores this process may actually use: min(affinity, cgroup quota); the GPU boxes expose 256 logical CPUs
    but give a one-GPU job a 16-CPU share, and oversubscribed oneDNN threads run far slower than 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    env = os.environ.get("CDX_CPU_THREADS")
    return int(env) if env else min(n, 16)


def cpu_baseline(cfg, params, cond_cpu, torch, budget_s=25.0):
    """Stock-torch CPU oracle: seconds per UNet forward + update at batch 1, bounded sample."""
    import oracle
    ncores = host_cores()
    torch.set_num_threads(ncores)
    H = cfg["image_size"]
    x = oracle.sampler_ref.noise_ref(0, 0, 1, 1, (3, H, H))
    coefs = oracle.step_coefficients_ref(100, "ddim")
    c1 = cond_cpu[:1]
    times = []
    t_begin = time.perf_counter()
    for k, (t, ca, cb, cx, c0, ce, sigma) in enumerate(coefs):
        t0 = time.perf_counter()
        eps = oracle.unet_forward_ref(cfg, params, x, torch.full((1,), t, dtype=torch.int64), c1)
        x = cx * x + c0 * (ca * x + cb * eps).clamp(-1, 1) + ce * eps
        times.append(time.perf_counter() - t0)
        if k >= 1 and time.perf_counter() - t_begin > budget_s:
            break
    timed = times[1:] if len(times) > 1 else times    # first step = warm-up
    sec_per_step = sum(timed) / len(timed)
    return {"value": round(1.0 / (sec_per_step * 100), 6), "unit": "images/s", "cores": ncores, "kind": "port",
            "sample": f"1 image x {len(timed)} of 100 DDIM steps at 256x256 (1 warm-up step dropped; every step "
                      f"costs the same), scaled to 100 steps; {sec_per_step:.2f} s/step, torch {torch.__version__} CPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU (BASELINE.json configs[1]: 16)")
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--details", action="store_true", help="print the per-variant conv table to stderr")
    args = ap.parse_args()

    import torch
    import cdx

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or "RANK" in os.environ:     # launched by torch.distributed.run: RCCL for the barrier / max-reduce only
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    cfg, run = cdx.named_config(args.config)
    B = args.batch
    # cfg5: the unit of work of a step is a batch of 256^2 TILES (fp16); an image is (image/stride)^2 tiles
    tiles_per_image = 1
    if "image" in run:
        tiles_per_image = len(cdx.tile_origins(run["image"], cfg["image_size"], run["overlap"])) ** 2
    params = cdx.init_params(cfg, seed=0)
    net = cdx.UNet(cfg, params, device=f"cuda:{local}")
    sampler = cdx.Sampler(net, method=run["method"])
    first_image = rank * B                                  # weak scaling: B images per GPU
    sb = cdx.synthetic_batch(cfg, 0, first_image, B)
    cond = torch.from_numpy(sb["cond"]).cuda()
    plan = net.plan(B)

    # inputs resident before the timed region
    from cdx.unet import load_cond
    load_cond(plan, cfg, cond)
    cdx.ops.gauss_fill(plan.xin, cfg["in_channels"], 0, first_image, cdx.rng.STREAM_XT)
    coefs = cdx.step_coefficients(sampler.schedule, run["steps"], run["method"])
    st = torch.cuda.current_stream().cuda_stream
    upd = cdx._abi.DiffusionUpdateArgs()
    upd.x, upd.x_ld, upd.eps, upd.eps_ld = plan.xin.data_ptr(), plan.xin.shape[-1], plan.eps.data_ptr(), plan.eps.shape[-1]
    upd.batch, upd.hw, upd.channels = B, cfg["image_size"] ** 2, cfg["in_channels"]
    upd.clip_x0, upd.seed, upd.first_image = 1, 0, first_image

    def step(k):
        c = coefs[k % len(coefs)]
        plan.t.fill_(c.t)
        plan.run(st)
        upd.ca, upd.cb, upd.cx, upd.c0, upd.ce, upd.sigma = c.ca, c.cb, c.cx, c.c0, c.ce, c.sigma
        upd.noise_stream = cdx.rng.STREAM_STEP0 + k
        cdx._abi.call("diffusion_update_f32", upd, None, 0, st)

    for k in range(args.warmup):
        step(k)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        tmax = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    assert torch.isfinite(plan.xin).all(), "non-finite state after the timed region"

    ms_per_step = elapsed / args.steps * 1e3
    images_per_s = (B * world / tiles_per_image) / (run["steps"] * ms_per_step * 1e-3)
    fl = plan_flops(plan)
    total_flops = sum(fl.values())
    line = {
        "metric": ("decoded images/sec (whole node), 256x256 100-step DDIM" if args.config in ("cfg2", "cfg3") else
                   f"decoded images/sec (whole node), {args.config}: {cfg['image_size']}x{cfg['image_size']} {run['steps']}-step {run['method'].upper()}"),
        "value": round(images_per_s, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16" if cfg["dtype"] == "fp16" else "f32", "data": "synthetic",
        "config": {"workload": (f"BASELINE.json configs[1]: 256x256x3, 128-ch UNet (channel_mult 1,1,2,2,4,4; "
                                f"self-attn at 16^2), {run['steps']}-step {run['method'].upper()}, batch {B} per GPU, "
                                f"seeded random weights (cdx.init_params seed 0)") if args.config in ("cfg2", "cfg3") else
                               (f"BASELINE.json {args.config}: {cfg['image_size']}^2 x3, {cfg['base_channels']}-ch UNet, dtype "
                                f"{cfg['dtype']}, cond_mode {cfg['cond_mode']}, {run['steps']}-step {run['method'].upper()}, "
                                f"batch {B} per GPU" + (f" = tiles of a {run['image']}^2 image ({tiles_per_image} tiles/image, "
                                                       f"overlap {run['overlap']})" if tiles_per_image > 1 else "")),
                   "images_per_gpu": B, "global_batch": B * world, "image": f"{cfg['image_size']}x{cfg['image_size']}x3",
                   "sampler_steps_per_image": run["steps"], "parallelism": f"replica x{world} (no collective)"},
        "algorithmic_gflop_per_step": round(total_flops / 1e9, 1),
        "achieved_tflops_whole_step": round(total_flops / (ms_per_step * 1e-3) / 1e12, 2),
    }
    if rank == 0 and not args.no_roofline:
        roof, table = measure_dominant_kernel(plan, torch)
        line["roofline"] = roof
        if args.details:
            print(json.dumps({"conv_variants": table, "flops": fl}, indent=1), file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(dict(cfg, dtype="fp32"), params, torch.from_numpy(sb["cond"]), torch)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
