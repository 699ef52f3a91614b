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

The timed steps are Sampler.step() calls -- the same function Sampler.sample() loops over (no private copy of the
loop here); the shard arithmetic (which images this rank decodes) and the barrier / max-over-ranks timing come from
cdx.shard (ShardJob, timed_region), the driver the CPU gloo test exercises.  `sample_call` additionally times ONE
complete Sampler.sample() of this rank's shard (x_T draw, all steps, export) after the step loop.

`python bench.py --gpus N` with N > 1 and no launcher around it starts the N ranks itself (torch.distributed.run as a child
process, before anything touches the GPU), forwards rank 0's JSON line and exits with the child's status.

Extra objects on the JSON line:
  roofline     -- (`traffic`: HBM bytes per launch of that kernel measured in THIS run by two short rocprofv3 --pmc child runs; the
                  committed digest under profiles/ is the fallback and is kept beside it as `traffic_digest`)
                  the dominant kernel symbol (the conv instantiation carrying most FLOPs): `achieved` = ALGORITHMIC FLOPs of
                  its launches in one forward / their summed durations (HIP events on the launch stream); `frac` (=
                  `frac_algorithmic`) = achieved / dense peak of the MFMA instruction it issues.  The float32 path issues 3
                  fp16 MFMAs per product (`emulation_factor` 3: frac <= 1/3); `mfma_pipe_utilisation` = the MFMA FLOPs it
                  issues / the same peak.
  strict_f32   -- the same step loop on the f32-input MFMA kernels only (UNet(split=False): direct + Winograd
                  v_mfma_f32_32x32x2_f32), a few steps, timed in the same run: the conservative float32 number.
  cpu_baseline -- the stock-torch CPU oracle (oracle/, kind "port": the reference ships no sampler) timed
                  on this host's cores on a bounded sample of the same workload (rank 0, N = 1 only): image 0 of the job, all of
                  its steps when they fit --cpu-budget (cfg2: 100 steps, ~60-75 s on 16 cores).
  parity       -- the metric's second half ("PSNR delta vs ref"): the HIP decode of image 0 (from the Sampler.sample call timed
                  as sample_call) against the oracle's decode of the same image from the cpu_baseline leg: PSNR(hip, oracle),
                  |PSNR(hip, target) - PSNR(oracle, target)|, max abs error, steps compared.  bench.py exits non-zero when
                  PSNR(hip, oracle) < 80 dB or the delta exceeds 0.01 dB (--no-parity-gate: report only).
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
    name = cdx._abi.TILE_NAMES[tile]
    if name.startswith("split") and (a.flags & cdx._abi.CONV_UPSAMPLE2X) and a.wpacked_split_up and a.ksize == 3 and a.win >= 32 and not a.residual:
        name = "split_up4x(2x2)"      # four 2x2 phase launches on the low-resolution source (conv_split.hip): 4/9 of the 3x3 layer's MFMAs
    return (a.ksize, a.stride, logtw, name)


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


def csrc_sha16() -> str:
    """Hash of the kernel sources this process runs (csrc/*.hip, *.h, include/cdx.h): stamps PMC digests in profiles/ so
    that a traffic figure measured on another build is not passed off as this build's."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "conditional-diffusion-model-for-compression_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))) + [os.path.join(ROOT, "include", "cdx.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def kernel_model(variant) -> dict:
    """What the kernel behind a conv variant executes on the matrix pipe, per ALGORITHMIC (direct-convolution) FLOP."""
    tile = variant[3]
    if tile.startswith("wino"):
        # Winograd F(2x2,3x3): 16 transform-domain multiply-adds per 2x2 output tile where the direct algorithm needs 36
        return {"name": "conv_wino8_kernel<WinoCfg<2,0>> (Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32)",
                "executed_per_algorithmic": 1.0 / 2.25, "peak": FP32_MFMA_PEAK_TFLOPS, "wino": True, "flops_per_mfma_cycle": 64.0,
                "pmc_pattern": "conv_wino8_kernel"}
    if tile.startswith("split_up"):
        return {"name": "conv16_ws_kernel<Conv16Cfg<KS = 2, ..., SPLIT, WS>> x 4 phases (3x3 after nearest-2x upsampling as four 2x2 convolutions on the low-resolution source)",
                "executed_per_algorithmic": 3.0 * 4.0 / 9.0, "peak": FP16_MFMA_PEAK_TFLOPS, "wino": False, "flops_per_mfma_cycle": 1024.0,
                "pmc_pattern": "Conv16Cfg<2, 1, 5, 4, 2, 0, 1, 1, 0, 1>"}
    if tile.startswith("split"):
        # float32 product on the fp16 matrix pipe: hi*hi + lo*hi + hi*lo = 3 fp16 MFMA FLOPs per algorithmic FLOP
        return {"name": ("conv16_ws_kernel<Conv16Cfg<..., SPLIT, WS>> (f32 conv as 3 x v_mfma_f32_32x32x16_f16 on hi|lo split operands; 4 MFMA + 4 producer waves)"
                         if variant[:3] == (3, 1, 5) else "conv16_kernel<Conv16Cfg<..., SPLIT>> (f32 conv as 3 x v_mfma_f32_32x32x16_f16 on hi|lo split operands)"),
                "executed_per_algorithmic": 3.0, "peak": FP16_MFMA_PEAK_TFLOPS, "wino": False, "flops_per_mfma_cycle": 1024.0,
                # rocprofv3 kernel-name substring of this variant (every staging mode STG of it): KS, ST, log2TW, MT, PF, ABL, SPLIT, DB, BF
                # (3x3 stride 1 at >= 32 px: the wave-specialised 8-wave workgroup, WS = 1 -- conv16_ws_kernel)
                "pmc_pattern": ("Conv16Cfg<3, 1, 4, 4, 3, 0, 1, 1, 0, 1>" if variant[:3] == (3, 1, 5) else      # (8 x 16-pixel tile: template LOGTW = 4, MT = 4)
                                "Conv16Cfg<3, 1, 4, 2, 3, 0, 1, 1, 0, 1>" if variant[:3] == (3, 1, 4) else                       # (16 px wide: wave-specialised 64-pixel tile)
                                "Conv16Cfg<%d, %d, %d, %d, 3, 0, 1, %d, 0, 0>" % (variant[0], variant[1], variant[2], 2 if (variant[1] == 2 or variant[2] < 5) else 4, 0 if variant[1] == 2 else 1))}
    if tile.startswith("f16"):
        return {"name": "conv16_kernel<Conv16Cfg> (v_mfma_f32_32x32x16_f16)", "executed_per_algorithmic": 1.0,
                "peak": FP16_MFMA_PEAK_TFLOPS, "wino": False, "flops_per_mfma_cycle": 1024.0,
                "pmc_pattern": ("Conv16Cfg<3, 1, 4, 4, 3, 0, 0, 1, " if variant[:3] == (3, 1, 5) else
                                "Conv16Cfg<%d, %d, %d, %d, 3, 0, 0, 1, " % (variant[0], variant[1], variant[2], 2 if (variant[1] == 2 or variant[2] in (3, 4)) else 4))}
    return {"name": "conv_kernel<ConvCfg> (direct implicit GEMM on v_mfma_f32_32x32x2_f32)", "executed_per_algorithmic": 1.0,
            "peak": FP32_MFMA_PEAK_TFLOPS, "wino": False, "flops_per_mfma_cycle": 64.0, "pmc_pattern": "conv_kernel<cdx::ConvCfg<%d, %d, %d, " % variant[:3]}


def find_traffic_digest(pattern, sha, workload, directory=None):
    """The committed PMC traffic digest that may be quoted for this run: same kernel (name pattern), same build of the kernel
    sources (csrc hash) and same workload (a digest is an average over the launches of ONE bench command; digests without a
    "config" field predate the check and were taken on cfg2).  Returns (file name, digest, None) or (None, None, description of
    the newest digest of the same kernel and workload from ANOTHER build, or None)."""
    directory = directory or os.path.join(ROOT, "profiles")
    stale = None
    for name in sorted(os.listdir(directory), reverse=True):
        if "_traffic" not in name or not name.endswith(".json"):
            continue
        try:
            tr = json.load(open(os.path.join(directory, name)))
        except (OSError, ValueError):
            continue
        if tr.get("pattern") != pattern or tr.get("config", "cfg2") != workload:
            continue
        if tr.get("csrc_sha16") == sha:
            return name, tr, None
        if stale is None:
            stale = {"file": f"profiles/{name}", "csrc_sha16": tr.get("csrc_sha16"), "hbm_bytes_per_launch": round(tr["hbm_bytes_per_launch"])}
    return None, None, stale


def live_traffic(pattern: str, workload: str, extra_args=(), timeout_s: float = 150.0):
    """HBM bytes per launch and matrix-pipe busy fraction of the kernel `pattern`, MEASURED IN THIS RUN: three rocprofv3 --pmc passes
    (FETCH_SIZE, WRITE_SIZE, then SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE: separate runs, as MI355X_MICROARCH.md's HBM section
    prescribes) over a short child run of this same bench.py (2 steps of the same workload, nothing else), started as ordinary child
    processes of this one (no exec), each in its own process group with a hard time limit.
    bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies a 128-byte read request at 64 B; rocprofv3 reports KiB).
    Returns ((bytes per launch, busy dict) | None, description)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    means = {}
    with tempfile.TemporaryDirectory(dir="/tmp", prefix="cdx_pmc_") as d:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"):
            out = os.path.join(d, ctr.split()[0])
            cmd = [exe, "--kernel-trace", "--pmc", *ctr.split(), "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--config", workload, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-sample-call",
                   "--no-strict-f32", "--no-live-traffic"] + list(extra_args)
            try:
                pr = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                      start_new_session=True)
                try:
                    rc = pr.wait(timeout=timeout_s)
                except subprocess.TimeoutExpired:
                    os.killpg(pr.pid, signal.SIGKILL)          # the profiler AND the bench child under it
                    pr.wait()
                    return None, f"{ctr} pass exceeded {timeout_s:.0f} s"
                if rc != 0:
                    return None, f"{ctr} pass exited with status {rc}"
            except OSError as e:
                return None, f"{ctr} pass could not start: {e}"
            for name in ctr.split():
                vals = []
                for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                    with open(f, newline="") as fh:
                        for r in csv.DictReader(fh):
                            if r["Counter_Name"] == name and pattern in r["Kernel_Name"]:
                                vals.append(float(r["Counter_Value"]))
                if not vals:
                    return None, f"{name} pass: no launch of the kernel in the counter output"
                means[name] = (sum(vals) / len(vals), len(vals))
    nbytes = 1024.0 * (2.0 * means["FETCH_SIZE"][0] + means["WRITE_SIZE"][0])
    # matrix pipe busy: SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's SIMDs) against the cycles the chip offered during the launch,
    # GRBM_GUI_ACTIVE (rocprofv3 reports the SUM over the 8 XCDs) / 8 x 256 CUs x 4 SIMDs
    busy = {"mfma_busy_cycles_per_launch": means["SQ_VALU_MFMA_BUSY_CYCLES"][0],
            "mfma_busy_frac": means["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (means["GRBM_GUI_ACTIVE"][0] / 8.0 * 256 * 4)}
    return (nbytes, busy), (f"live: rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, one run each) over "
                            f"`bench.py --config {workload} --steps 2 --warmup 1` as child processes of this run; mean over {means['FETCH_SIZE'][1]} launches; traffic = 2 x FETCH + WRITE")


def measure_dominant_kernel(plan, torch, reps=3, workload="cfg2"):
    """Per-launch HIP-event timing of every conv launch of one forward; returns the roofline object for the
    kernel symbol with the most FLOPs, plus a per-variant table."""
    import ctypes
    st = torch.cuda.current_stream()
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
    km = kernel_model(dom)
    algorithmic = d["flops"] / (d["ms"] * 1e-3) / 1e12
    executed = algorithmic * km["executed_per_algorithmic"]
    peak = km["peak"]
    ef = km["executed_per_algorithmic"]
    # achieved = ALGORITHMIC FLOPs of the launches / their measured time; frac = achieved / peak of the pipe the
    # kernel runs on.  A kernel that emulates float32 products with 3 fp16 MFMAs (emulation_factor 3) can reach at most 1/3 of
    # that peak in algorithmic terms; mfma_pipe_utilisation = executed MFMA FLOPs / peak says how busy it keeps the pipe.
    roof = {"roofline_schema": 2,    # 1 (rounds 1-2): frac = EXECUTED MFMA FLOPs / peak; 2 (round 3 on): frac = ALGORITHMIC FLOPs / peak
            "bound": "mfma", "achieved": round(algorithmic, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(algorithmic / peak, 4), "traffic": None,
            "frac_algorithmic": round(algorithmic / peak, 4),
            "emulation_factor": ef, "mfma_pipe_utilisation": round(executed / peak, 4), "executed_tflops": round(executed, 2),
            "kernel": km["name"] + " [ksize %d stride %d layers >= 2^%d px wide, tile id %s; 3x3 stride 1 at >= 32 px runs 8 x 16-pixel tiles]" % dom,
            "flop_accounting": "achieved / frac / frac_algorithmic = algorithmic (direct-convolution) FLOPs 2*Cin*Cout*k*k*H*W*B per second "
                               "against the dense peak of the MFMA instruction the kernel issues; executed_tflops / mfma_pipe_utilisation = "
                               "the MFMA FLOPs it actually issues (emulation_factor x algorithmic) against the same peak"
                               + ("; Winograd F(2x2,3x3) executes 1/2.25 of the direct-convolution FLOPs" if km["wino"] else ""),
            "algorithmic_tflops": round(algorithmic, 2), "algorithmic_over_f32_mfma_peak": round(algorithmic / FP32_MFMA_PEAK_TFLOPS, 4),
            "avg_launch_ms": round(d["ms"] / d["launches"], 4), "launches_per_forward": d["launches"] // reps,
            "flop_share_of_forward": round(d["flops"] / sum(v["flops"] for v in table.values()), 4),
            "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
            # MFMA instructions x issue cycles, summed over all waves of an average launch (unpadded shapes): the value
            # SQ_VALU_MFMA_BUSY_CYCLES should report per launch (tools/pmc_digest.py mfma ... unit check)
            "mfma_cycles_per_launch_expected": round(d["flops"] * km["executed_per_algorithmic"] / d["launches"] / km["flops_per_mfma_cycle"]),
            "hbm_GBps_at_algorithmic_bytes": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1)}
    # HBM bytes per launch of this kernel: PMC counters cannot be read from inside the process, so the figure comes from
    # the committed rocprofv3 --pmc passes over this same command -- accepted only if they were taken on THIS build of
    # the kernels (csrc hash) and for THIS kernel; otherwise traffic stays null and the stale digest is named.
    sha = csrc_sha16()
    roof["csrc_sha16"] = sha
    roof["pmc_pattern"] = km["pmc_pattern"]
    name, tr, stale = find_traffic_digest(km["pmc_pattern"], sha, workload)
    if tr is not None:
        roof["traffic"] = round(tr["hbm_bytes_per_launch"])
        roof["traffic_over_algorithmic"] = round(tr["hbm_bytes_per_launch"] / roof["algorithmic_bytes_per_launch"], 3)
        roof["traffic_source"] = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on this build: 2 x FETCH + WRITE)"
    elif stale is not None:
        roof["traffic_from_other_build"] = stale
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


def cpu_baseline(cfg, run, params, cond_cpu, torch, budget_s=300.0):
    """Stock-torch CPU oracle on image 0 of the job (same cond, x_T stream and per-step noise streams as the GPU's image 0):
    seconds per UNet forward + update at batch 1, on as many of the run's steps as fit into budget_s (default: all of them for
    cfg2: ~0.6 s/step).  Returns (the cpu_baseline object, the oracle's state after the last step it ran [1,3,H,H] -- clamped if
    that was the run's final step --, the number of steps it ran)."""
    import oracle
    ncores = host_cores()
    torch.set_num_threads(ncores)
    H, S = cfg["image_size"], run["steps"]
    x = oracle.sampler_ref.noise_ref(0, 0, 1, 1, (3, H, H))
    coefs = oracle.step_coefficients_ref(S, run["method"])
    c1 = cond_cpu[:1]
    times = []
    t_begin = time.perf_counter()
    for k, (t, ca, cb, cx, c0, ce, sigma) in enumerate(coefs):
        t0 = time.perf_counter()
        eps = oracle.unet_forward_ref(cfg, params, x, torch.full((1,), t, dtype=torch.int64), c1)
        x = cx * x + c0 * (ca * x + cb * eps).clamp(-1, 1) + ce * eps
        if sigma != 0.0:
            x = x + sigma * oracle.sampler_ref.noise_ref(0, 0, 1, 16 + k, (3, H, H))
        times.append(time.perf_counter() - t0)
        if k >= 1 and time.perf_counter() - t_begin > budget_s:
            break
    done = len(times)
    if done == S:
        x = x.clamp(-1.0, 1.0)
    timed = times[1:] if len(times) > 1 else times    # first step = warm-up
    sec_per_step = sum(timed) / len(timed)
    tiles = 1
    if "image" in run:
        import cdx
        tiles = len(cdx.tile_origins(run["image"], H, run["overlap"])) ** 2
    return ({"value": round(1.0 / (sec_per_step * S * tiles), 6), "unit": "images/s", "cores": ncores, "kind": "port",
             "sample": f"1 image x {len(timed)} of {S} {run['method'].upper()} steps at {H}x{H} (1 warm-up step dropped; every step "
                       f"costs the same), scaled to {S} steps" + (f" x {tiles} tiles per {run['image']}^2 image" if tiles > 1 else "")
                       + f"; {sec_per_step:.2f} s/step, float32, torch {torch.__version__} CPU"}, x, done)


PARITY_MIN_PSNR_DB = 80.0     # gate 2 (build-imposed): PSNR(hip, oracle), peak-to-peak 2.0
PARITY_MAX_DPSNR_DB = 0.01    # gate 1 (BASELINE.json north_star): |PSNR(hip, target) - PSNR(oracle, target)|


def psnr_db(a, b) -> float:
    import math
    mse = ((a.double() - b.double()) ** 2).mean().item()
    return float("inf") if mse == 0 else 10.0 * math.log10(4.0 / mse)


def parity_object(hip, oracle_x, target, steps_done, steps_total, how) -> dict:
    """The metric's second half ("PSNR delta vs ref", SURVEY.md 8d gates 1-2) for image 0: the HIP decode against the CPU oracle's
    decode of the same image (same cond, x_T, weights).  `complete` = the oracle ran every step (states compared are then the
    final clamped images; otherwise the x_t after `steps` reverse steps)."""
    hip, oracle_x, target = hip.detach().float().cpu(), oracle_x.float().cpu(), target.float().cpu()
    p_ho, p_ht, p_ot = psnr_db(hip, oracle_x), psnr_db(hip, target), psnr_db(oracle_x, target)
    complete = steps_done == steps_total
    ok = p_ho >= PARITY_MIN_PSNR_DB and (not complete or abs(p_ht - p_ot) <= PARITY_MAX_DPSNR_DB)
    return {"psnr_hip_vs_oracle_db": round(p_ho, 3), "psnr_delta_vs_target_db": round(abs(p_ht - p_ot), 7),
            "psnr_hip_vs_target_db": round(p_ht, 5), "psnr_oracle_vs_target_db": round(p_ot, 5),
            "max_abs_err": float((hip - oracle_x).abs().max().item()), "steps": steps_done, "of_steps": steps_total, "complete": complete,
            "image": 0, "hip_image_from": how, "oracle": "oracle.unet_forward_ref + update on the CPU, float32 (PARITY UNPINNED by the reference: it ships no sampler)",
            "gates": {"psnr_hip_vs_oracle_db_min": PARITY_MIN_PSNR_DB, "psnr_delta_vs_target_db_max": PARITY_MAX_DPSNR_DB}, "pass": bool(ok)}


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n: int, argv: list, entry: str | None = None) -> int:
    """Start n ranks of `entry` (default: this file) under torch.distributed.run as ONE child process and return its exit
    status.  The caller must not have touched the GPU (no exec, no fork of an initialised process: the child is a fresh
    interpreter); stdout / stderr are inherited, so rank 0's JSON line reaches the caller's stdout unchanged."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), entry or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def strict_f32_leg(cdx, torch, job, cfg, run, B, tiles_per_image, total_flops, tiled, nsteps=5):
    """The same step loop on the f32-input MFMA kernels only (UNet(split=False)): no fp16 operands anywhere, a few steps
    (every step costs the same), timed in this run so that the conservative float32 number is the driver's own."""
    net = cdx.UNet(cfg, job.params, device=job.device, split=False)
    smp = cdx.Sampler(net, method=run["method"])
    if tiled:
        conds, xts, _, _ = cdx.tiling.tile_batch(net, job.cond(job.lo, 1), run["overlap"], job.seed, job.lo)
        reps_ = -(-B // conds.shape[0])
        st = smp.begin(conds.repeat(reps_, 1, 1, 1)[:B].contiguous(), run["steps"], seed=job.seed, x_T=xts.repeat(reps_, 1, 1, 1)[:B].contiguous())
    else:
        st = smp.begin(job.cond(job.lo, B), run["steps"], seed=job.seed, first_image=job.lo)
    smp.step(st, 0)
    sec = cdx.timed_region(lambda: [smp.step(st, 1 + k) for k in range(nsteps)], None, torch.cuda.synchronize)
    assert torch.isfinite(st.plan.xin).all()
    ms = sec / nsteps * 1e3
    tf = total_flops / (ms * 1e-3) / 1e12
    out = {"images_per_s": round((B / tiles_per_image) / (run["steps"] * ms * 1e-3), 4), "ms_per_step": round(ms, 3), "steps": nsteps,
           "algorithmic_tflops_whole_step": round(tf, 2), "algorithmic_over_direct_f32_peak": round(tf / FP32_MFMA_PEAK_TFLOPS, 4), "peak": FP32_MFMA_PEAK_TFLOPS,
           "arithmetic": "float32 operands on v_mfma_f32_32x32x2_f32 (direct + Winograd F(2x2,3x3): algorithmic rate can exceed the direct-convolution peak)"}
    del smp, st, net
    torch.cuda.empty_cache()
    return out


def bind_gpu(torch, local: int) -> str:
    """One process per GPU: the rank whose LOCAL_RANK is l drives cuda:l (and nothing else)."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    return f"cuda:{local}"


def main(argv=None, make_sampler=None, dist_backend=None, bind_device=None):
    """make_sampler: test injection (tests/bench_entry_oracle.py runs this same function on CPU ranks with a stand-in sampler to
    exercise the launch, shard and timing path without a GPU); the product run passes none.  dist_backend overrides
    --dist-backend.  The control plane (barrier + max-over-ranks of one float64) runs over GLOO on CPU tensors by default, in
    the product run too: the data path has no collective (north_star: "no RCCL required"), and gloo is the backend the CPU
    tests exercise; `--dist-backend nccl` initialises RCCL for the same two calls instead.  bind_device (default: not injected):
    bind this rank to cuda:LOCAL_RANK (the 8-rank CPU rehearsal passes True with torch.cuda mocked to check the binding)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="images (cfg5: tiles) per GPU and sampler call; default: the config's per-GPU batch")
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--graph", action="store_true", help="replay each UNet forward as one hipGraph launch (Sampler(use_graph=True): host-bound sizes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=200.0, help="seconds of CPU-oracle work in the cpu_baseline / parity leg (cfg2's 100 steps need ~60-75 s on 16 cores)")
    ap.add_argument("--no-parity-gate", action="store_true", help="report the parity object but do not exit non-zero when it fails its gates")
    ap.add_argument("--dist-backend", default="gloo", choices=["gloo", "nccl"], help="control plane of a multi-rank run (barrier + max-reduce only)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sample-call", action="store_true", help="skip the extra timing of one complete Sampler.sample()")
    ap.add_argument("--no-strict-f32", action="store_true", help="skip the extra step loop on the f32-input MFMA kernels (strict_f32)")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed PMC digest only (default: measured in this run by two "
                                                                   "short rocprofv3 --pmc child runs, ~1 min; the digest is the fallback)")
    ap.add_argument("--dtype", default=None, choices=["fp32", "fp16", "bf16"], help="override the config's storage dtype (e.g. cfg2 in fp16 / bf16)")
    ap.add_argument("--no-split", action="store_true", help="A/B: float32 layers on the f32-input MFMA kernels only (no fp16 hi|lo split tile)")
    ap.add_argument("--details", action="store_true", help="print the per-variant conv table to stderr")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the ranks ourselves, BEFORE importing torch or touching the GPU in this process
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else argv))

    import torch
    import cdx

    injected = make_sampler is not None
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if bind_device if bind_device is not None else not injected:
        bind_gpu(torch, local)
    dist = None
    backend = dist_backend or args.dist_backend
    if world > 1 or "RANK" in os.environ:     # launched by torch.distributed.run: a process group for the barrier / max-reduce only
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo" and os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            # single node: gloo's pairwise connections go over loopback (its default picks the interface the HOSTNAME resolves to,
            # and a container's hostname may not resolve)
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if backend == "nccl" and not injected:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    # This rank's shard of the job: weak scaling, B units per GPU (cdx.shard owns the rank arithmetic).
    cfg, run = cdx.named_config(args.config)
    if args.dtype:
        cfg = cdx.unet_config(**dict(cfg, dtype=args.dtype))
    tiled = "image" in run
    per_gpu = cdx.shard.IMAGES_PER_CALL[args.config]          # images of the job per GPU (BASELINE.json: 16 / 16 / 8 / 8)
    B = args.batch or (16 if tiled else per_gpu)              # units per sampler step and GPU: images, or (cfg5) TILES
    tiles_per_image = len(cdx.tile_origins(run["image"], cfg["image_size"], run["overlap"])) ** 2 if tiled else 1
    images_per_gpu = per_gpu if tiled else B      # cfg5: the step loop runs B TILES per step; sample_call decodes the rank's whole images
    device = None if injected else f"cuda:{local}"
    job = cdx.shard.ShardJob(images_per_gpu * world, args.config, rank=rank, world=world, device=device,
                             images_per_call=1 if tiled else images_per_gpu, config=(cfg, run), unet_kw={"split": False} if args.no_split else None,
                             make_sampler=make_sampler)
    sampler = job.sampler
    net = getattr(sampler, "unet", None)
    sampler.use_graph = bool(args.graph)
    sync = None if injected else torch.cuda.synchronize

    # inputs resident in HBM before the timed region: Sampler.begin() loads cond and draws x_T
    if tiled:
        tile_batch = getattr(sampler, "tile_batch", None) or (lambda cond, overlap, seed, first: cdx.tiling.tile_batch(net, cond, overlap, seed, first))
        conds, xts, _, _ = tile_batch(job.cond(job.lo, 1), run["overlap"], job.seed, job.lo)
        reps_ = -(-B // conds.shape[0])
        state = sampler.begin(conds.repeat(reps_, 1, 1, 1)[:B].contiguous(), run["steps"], seed=job.seed,
                              x_T=xts.repeat(reps_, 1, 1, 1)[:B].contiguous())
    else:
        state = sampler.begin(job.cond(job.lo, B), run["steps"], seed=job.seed, first_image=job.lo)
    plan = state.plan

    def steps(k0, n, smp=sampler, st=state):
        for k in range(k0, k0 + n):
            smp.step(st, k % run["steps"])      # the function Sampler.sample() loops over

    steps(0, args.warmup)
    elapsed = cdx.timed_region(lambda: steps(args.warmup, args.steps), dist, sync)
    assert torch.isfinite(plan.xin).all(), "non-finite state after the timed region"

    ms_per_step = elapsed / args.steps * 1e3
    images_per_s = (B * world / tiles_per_image) / (run["steps"] * ms_per_step * 1e-3)
    fl = plan_flops(plan) if not injected else {"conv": 0.0}
    total_flops = sum(fl.values())
    line = {
        "metric": ("decoded images/sec (whole node), 256x256 100-step DDIM" if args.config in ("cfg2", "cfg3") else
                   f"decoded images/sec (whole node), {args.config}: {run.get('image', cfg['image_size'])}^2 {run['steps']}-step {run['method'].upper()}"),
        "value": round(images_per_s, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp16": "f16", "bf16": "bf16"}.get(cfg["dtype"], "f32"), "data": "synthetic",
        "arithmetic": ("%s storage, v_mfma_f32_32x32x16_%s, float32 accumulate" % (cfg["dtype"], "bf16" if cfg["dtype"] == "bf16" else "f16") if cfg["dtype"] != "fp32" else
                       "float32 storage and results; layers >= 8 px wide: float32 products as 3 fp16 MFMAs on hi|lo split operands, "
                       "float32 accumulate (CDX_TILE_SPLIT); other layers: v_mfma_f32_32x32x2_f32" if not args.no_split else
                       "float32 storage, v_mfma_f32_32x32x2_f32 (direct + Winograd F(2x2,3x3))"),
        "config": {"workload": (f"BASELINE.json configs[1]: 256x256x3, 128-ch UNet (channel_mult 1,1,2,2,4,4; "
                                f"self-attn at 16^2), {run['steps']}-step {run['method'].upper()}, batch {B} per GPU, "
                                f"seeded random weights (cdx.init_params seed 0)") if args.config in ("cfg2", "cfg3") else
                               (f"BASELINE.json {args.config}: {cfg['image_size']}^2 x3, {cfg['base_channels']}-ch UNet, dtype "
                                f"{cfg['dtype']}, cond_mode {cfg['cond_mode']}, {run['steps']}-step {run['method'].upper()}, "
                                f"batch {B} per GPU" + (f" = tiles of a {run['image']}^2 image ({tiles_per_image} tiles/image, "
                                                       f"overlap {run['overlap']})" if tiled else "")),
                   "images_per_gpu": images_per_gpu, "global_batch": images_per_gpu * world,
                   "units_per_step_per_gpu": B, "unit_of_a_step": "tile" if tiled else "image",
                   "image": f"{run.get('image', cfg['image_size'])}x{run.get('image', cfg['image_size'])}x3",
                   "sampler_steps_per_image": run["steps"], "parallelism": f"replica x{world} (no collective)",
                   "timed": "Sampler.step() x steps (cdx.sampler), shard + timing from cdx.shard.ShardJob / timed_region"
                            + (", UNet forward replayed as one hipGraph per step" if args.graph else "")},
        "algorithmic_gflop_per_step": round(total_flops / 1e9, 1),
        "algorithmic_tflops_whole_step": round(total_flops / (ms_per_step * 1e-3) / 1e12, 2),
    }
    out = {}
    if not args.no_sample_call:
        # ONE complete decode of this rank's shard through the API entry point itself: Sampler.sample (cfg5:
        # Sampler.sample_tiled of each of the rank's whole images, 16 tiles per call) -- x_T draw, every step, export; max over ranks.
        sec = cdx.timed_region(lambda: out.update(job.decode()), dist, sync)
        n_img = images_per_gpu * world
        line["sample_call"] = {"entry": "Sampler.sample_tiled(cond, steps)" if tiled else "Sampler.sample(cond, steps)",
                               "images": n_img, "steps": run["steps"], "seconds": round(sec, 4),
                               "images_per_s": round(n_img / sec, 4),
                               "ms_per_step_equiv": round(sec * 1e3 / (run["steps"] * (-(-tiles_per_image // 16) * images_per_gpu if tiled else 1)), 3)}
        assert all(torch.isfinite(v).all() for v in out.values())
    if injected:
        line["backend"] = "injected sampler (test of the launch / shard / timing path; not a measurement of the product)"
    if rank == 0 and not args.no_roofline and not injected:
        roof, table = measure_dominant_kernel(plan, torch, workload=args.config)
        line["roofline"] = roof
        if args.details:
            print(json.dumps({"conv_variants": table, "flops": fl}, indent=1), file=sys.stderr)
    if rank == 0 and world == 1 and not injected and not args.no_strict_f32 and not args.no_split and cfg["dtype"] == "fp32":
        line["strict_f32"] = strict_f32_leg(cdx, torch, job, cfg, run, B, tiles_per_image, total_flops, tiled)
    under_profiler = "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)      # (no profiler inside a profiler)
    if rank == 0 and world == 1 and not injected and "roofline" in line and not args.no_live_traffic and not under_profiler:
        # HBM traffic of the dominant kernel measured in THIS run (the digest under profiles/ is the builder's number)
        roof = line["roofline"]
        try:
            extra = (["--dtype", args.dtype] if args.dtype else []) + (["--no-split"] if args.no_split else []) + (["--batch", str(args.batch)] if args.batch else [])
            nbytes, how = live_traffic(roof["pmc_pattern"], args.config, extra)
        except Exception as e:      # never lose the line to the extra measurement
            nbytes, how = None, f"failed: {e!r}"
        if nbytes is not None:
            nbytes, busy = nbytes
            roof["mfma_busy_frac"] = round(busy["mfma_busy_frac"], 4)
            # unit check: the counter must equal the kernel's MFMA instruction count x issue cycles (computed from the launch list)
            roof["mfma_busy_counter_over_expected"] = round(busy["mfma_busy_cycles_per_launch"] / roof["mfma_cycles_per_launch_expected"], 4)
            if roof.get("traffic") is not None:
                roof["traffic_digest"] = {"bytes": roof["traffic"], "source": roof.get("traffic_source")}
            roof["traffic"] = round(nbytes)
            roof["traffic_over_algorithmic"] = round(nbytes / roof["algorithmic_bytes_per_launch"], 3)
            roof["traffic_source"] = how
        else:
            roof["live_traffic_unavailable"] = how
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not injected:
        c1 = job.cond(0, 1).cpu()
        if tiled:
            c1 = c1[..., :cfg["image_size"] // 16, :cfg["image_size"] // 16].contiguous()
        line["cpu_baseline"], x_cpu, done = cpu_baseline(dict(cfg, dtype="fp32"), run, job.params, c1, torch, budget_s=args.cpu_budget)
        if tiled:
            line["parity"] = {"skipped": "tiled configuration: the CPU leg decodes one tile with its own x_T; the whole-image parity "
                                         "check is tests/test_configs_gpu.py (25 fp16 tiles vs the oracle blend)"}
        else:
            # the metric's second half: the HIP decode of image 0 against the oracle's decode of the same image.  If the oracle ran all
            # steps and sample_call decoded this shard, the HIP image is image 0 of THAT Sampler.sample call; otherwise image 0 is
            # decoded again alone with a trace (bit-identical to its in-batch decode: tests/test_e2e_gpu.py sharding invariance).
            S = run["steps"]
            if done == S and 0 in out:
                hip, how = out[0][None], "image 0 of the Sampler.sample(cond, steps) call timed as sample_call"
            else:
                trace = []
                fin = sampler.sample(job.cond(0, 1), S, seed=job.seed, first_image=0, trace=trace)
                hip = fin if done == S else trace[done - 1]
                how = f"Sampler.sample(cond[0:1], {S}, trace=...) state after step {done}"
            line["parity"] = parity_object(hip, x_cpu, torch.from_numpy(job.inputs(0, 1)["target"]), done, S, how)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)
        par = line.get("parity")
        if par and par.get("pass") is False and not args.no_parity_gate:
            raise SystemExit(f"parity gate failed: {par}")


if __name__ == "__main__":
    main()
