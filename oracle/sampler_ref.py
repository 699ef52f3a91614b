"""Oracle reverse-diffusion loop (S1-S4 of SURVEY.md section 8a).  TEST INFRASTRUCTURE ONLY.

Independent restatement of the schedule, the DDIM (eta = 0) / DDPM ancestral update and the
sampling loop with stock torch on the CPU.  PARITY UNPINNED: the reference snapshot holds no
sampler (README.md 0 bytes); equations: Ho et al. 2020 eq. 6-7 / 11, Song et al. 2021 eq. 12.

    DDIM:  x0h = clip((x_t - sqrt(1-ab_t) eps) / sqrt(ab_t));  x_prev = sqrt(ab_p) x0h + sqrt(1-ab_p) eps
    DDPM:  x0h as above;  mu = sqrt(ab_p) b/(1-ab_t) x0h + sqrt(a)(1-ab_p)/(1-ab_t) x_t,
           a = ab_t/ab_p, b = 1-a;  x_prev = mu + sqrt(b (1-ab_p)/(1-ab_t)) z,  z = 0 at the last step
    tau_i = floor(i T / S);  x_T ~ N(0, I) from stream (seed, image, 1); step-k noise from stream
    (seed, image, 16 + k);  returns clamp(x_0, -1, 1).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .rng_ref import normal_ref, stream_key_ref
from .unet_ref import unet_forward_ref


def make_schedule_ref(T=1000, beta_start=1e-4, beta_end=2e-2):
    betas = [beta_start + (beta_end - beta_start) * i / (T - 1) for i in range(T)]
    ab, acc = [], 1.0
    for b in betas:
        acc *= (1.0 - b)
        ab.append(acc)
    return betas, ab


def step_coefficients_ref(steps: int, method: str, T=1000, beta_start=1e-4, beta_end=2e-2):
    """List of (t, ca, cb, cx, c0, ce, sigma) in execution order."""
    _, ab = make_schedule_ref(T, beta_start, beta_end)
    taus = [(i * T) // steps for i in range(steps)]
    out = []
    for i in reversed(range(steps)):
        t = taus[i]
        ab_t = ab[t]
        ab_p = ab[taus[i - 1]] if i > 0 else 1.0
        ca = 1.0 / math.sqrt(ab_t)
        cb = -math.sqrt(1.0 - ab_t) / math.sqrt(ab_t)
        if method == "ddim":
            out.append((t, ca, cb, 0.0, math.sqrt(ab_p), math.sqrt(1.0 - ab_p), 0.0))
        else:
            a = ab_t / ab_p
            b = 1.0 - a
            out.append((t, ca, cb, math.sqrt(a) * (1.0 - ab_p) / (1.0 - ab_t),
                        math.sqrt(ab_p) * b / (1.0 - ab_t), 0.0,
                        math.sqrt(b * (1.0 - ab_p) / (1.0 - ab_t)) if i > 0 else 0.0))
    return out


def noise_ref(seed: int, first_image: int, count: int, stream: int, shape) -> torch.Tensor:
    n = int(np.prod(shape))
    return torch.from_numpy(np.stack([
        normal_ref(stream_key_ref(seed, first_image + k, stream), n).reshape(shape) for k in range(count)]))


def sample_ref(cfg: dict, params: dict, cond: torch.Tensor, steps: int, *, seed: int = 0,
               method: str = "ddim", clip_x0: bool = True, first_image: int = 0,
               dtype=torch.float32, schedule: dict | None = None, trace: list | None = None,
               x_T: torch.Tensor | None = None) -> torch.Tensor:
    sc = dict(T=1000, beta_start=1e-4, beta_end=2e-2)
    sc.update({k: v for k, v in (schedule or {}).items() if k in sc})
    B = cond.shape[0]
    H = cfg["image_size"]
    shape = (cfg["in_channels"], H, H)
    x = (noise_ref(seed, first_image, B, 1, shape) if x_T is None else x_T).to(dtype)
    for k, (t, ca, cb, cx, c0, ce, sigma) in enumerate(step_coefficients_ref(steps, method, **sc)):
        tt = torch.full((B,), t, dtype=torch.int64)
        eps = unet_forward_ref(cfg, params, x, tt, cond, dtype=dtype)
        x0h = ca * x + cb * eps
        if clip_x0:
            x0h = x0h.clamp(-1.0, 1.0)
        x = cx * x + c0 * x0h + ce * eps
        if sigma != 0.0:
            x = x + sigma * noise_ref(seed, first_image, B, 16 + k, shape).to(dtype)
        if trace is not None:
            trace.append(x.clone())
    return x.clamp(-1.0, 1.0)
