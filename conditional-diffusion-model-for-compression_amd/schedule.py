"""Noise schedule tables and per-step update coefficients (host, float64).

S1 / S3 of SURVEY.md section 8(a).  No reference file exists for these (reference
snapshot is empty: README.md 0 bytes); equations are Ho et al. 2020 (DDPM) and
Song et al. 2021 (DDIM, eta = 0), fixed by SURVEY.md Appendix A.

One reverse step is always evaluated in the form

    x0h    = clamp(ca * x_t + cb * eps, -1, 1)          (clamp skipped if clip_x0 is False)
    x_prev = cx * x_t + c0 * x0h + ce * eps + sigma * z

so the DDIM and DDPM updates are one device kernel with six scalars.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from .config import SCHEDULE_DEFAULTS


def make_schedule(schedule_cfg: dict | None = None) -> dict:
    cfg = dict(SCHEDULE_DEFAULTS, **(schedule_cfg or {}))
    T = int(cfg["T"])
    if cfg["kind"] != "linear":
        raise ValueError(f"schedule kind {cfg['kind']!r} not supported")
    betas = np.linspace(float(cfg["beta_start"]), float(cfg["beta_end"]), T, dtype=np.float64)
    alphas_cumprod = np.cumprod(1.0 - betas)
    return dict(T=T, betas=betas, alphas_cumprod=alphas_cumprod)


def timestep_subsequence(T: int, steps: int) -> np.ndarray:
    """tau_i = floor(i * T / steps), i = 0..steps-1 (ascending)."""
    if not 1 <= steps <= T:
        raise ValueError(f"steps must be in [1, {T}], got {steps}")
    return (np.arange(steps, dtype=np.int64) * T) // steps


@dataclass(frozen=True)
class StepCoef:
    t: int          # timestep fed to the UNet
    ca: float       # x0h = clamp(ca*x + cb*eps)
    cb: float
    cx: float       # x_prev = cx*x + c0*x0h + ce*eps + sigma*z
    c0: float
    ce: float
    sigma: float


def step_coefficients(schedule: dict, steps: int, method: str) -> list[StepCoef]:
    """Coefficients for the reverse loop, in execution order (largest t first)."""
    ab = schedule["alphas_cumprod"]
    taus = timestep_subsequence(schedule["T"], steps)
    out = []
    for i in range(steps - 1, -1, -1):
        t = int(taus[i])
        ab_t = float(ab[t])
        ab_p = float(ab[int(taus[i - 1])]) if i > 0 else 1.0
        ca = 1.0 / math.sqrt(ab_t)
        cb = -math.sqrt(1.0 - ab_t) / math.sqrt(ab_t)
        if method == "ddim":
            out.append(StepCoef(t, ca, cb, 0.0, math.sqrt(ab_p), math.sqrt(1.0 - ab_p), 0.0))
        elif method == "ddpm":
            a_t = ab_t / ab_p                 # respaced alpha
            b_t = 1.0 - a_t
            c0 = math.sqrt(ab_p) * b_t / (1.0 - ab_t)
            cx = math.sqrt(a_t) * (1.0 - ab_p) / (1.0 - ab_t)
            var = b_t * (1.0 - ab_p) / (1.0 - ab_t)
            out.append(StepCoef(t, ca, cb, cx, c0, 0.0, math.sqrt(var) if i > 0 else 0.0))
        else:
            raise ValueError(f"method must be 'ddim' or 'ddpm', got {method!r}")
    return out
