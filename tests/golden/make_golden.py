#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle (stock torch, float32, this container).

The reference snapshot holds no golden vectors, tests or weights (README.md 0 bytes), so these pin the
BUILD's own contract: they make later rounds (and the GPU path) answer to the numbers produced here.
Inputs and weights are regenerated from seeds (cdx.init_params / cdx.synthetic_batch), never stored.

  cfg1_ddim50.npz   : BASELINE.json configs[0] -- 32x32x3, 64-ch UNet, 50 DDIM steps, batch 1, seed 0:
                      x0 [1,3,32,32] float32, plus x after steps 1, 10, 25 and the first UNet eps.
  tiny_ddpm.npz     : 16x16 32-ch UNet, 8 DDPM (ancestral) steps, batch 2, seed 5 (noise-injection path).
  rng.npz           : first 1024 normals / uniforms of three streams + stream keys.
  schedule.npz      : alphas_cumprod[1000] and the 100-step DDIM / 250-step DDPM coefficient tables.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cdx      # noqa: E402
import oracle   # noqa: E402

TINY = dict(image_size=16, base_channels=32, channel_mult=(1, 2), attn_resolutions=(8,), num_res_blocks=1)


def main():
    torch.set_num_threads(8)
    cfg, run = cdx.named_config("cfg1")
    params = cdx.init_params(cfg, seed=0)
    cond = torch.from_numpy(cdx.synthetic_batch(cfg, 0, 0, 1)["cond"])
    trace = []
    x0 = oracle.sample_ref(cfg, params, cond, run["steps"], seed=0, method="ddim", trace=trace)
    xT = oracle.sampler_ref.noise_ref(0, 0, 1, 1, (3, 32, 32))
    t0 = oracle.step_coefficients_ref(run["steps"], "ddim")[0][0]
    eps0 = oracle.unet_forward_ref(cfg, params, xT, torch.full((1,), t0, dtype=torch.int64), cond)
    np.savez_compressed(os.path.join(HERE, "cfg1_ddim50.npz"), x0=x0.numpy(), x_step1=trace[0].numpy(),
                        x_step10=trace[9].numpy(), x_step25=trace[24].numpy(), eps0=eps0.numpy(), t0=t0)

    tcfg = cdx.unet_config(**TINY)
    tparams = cdx.init_params(tcfg, seed=5, affine_jitter=0.1)
    tcond = torch.from_numpy(cdx.synthetic_batch(tcfg, 5, 0, 2)["cond"])
    tx0 = oracle.sample_ref(tcfg, tparams, tcond, 8, seed=5, method="ddpm")
    np.savez_compressed(os.path.join(HERE, "tiny_ddpm.npz"), x0=tx0.numpy())

    keys = [oracle.stream_key_ref(s, a, b) for (s, a, b) in [(0, 0, 1), (7, 3, 16), (123456789, 1 << 40, 0)]]
    np.savez_compressed(os.path.join(HERE, "rng.npz"), keys=np.array(keys, dtype=np.uint64),
                        normal=np.stack([oracle.normal_ref(k, 1024) for k in keys]),
                        uniform=np.stack([oracle.uniform_ref(k, 1024) for k in keys]))

    _, ab = oracle.make_schedule_ref()
    np.savez_compressed(os.path.join(HERE, "schedule.npz"), alphas_cumprod=np.array(ab),
                        ddim100=np.array(oracle.step_coefficients_ref(100, "ddim")),
                        ddpm250=np.array(oracle.step_coefficients_ref(250, "ddpm")))
    # (f4) bitstream side: one small CDXL container from the oracle's rANS encoder + the symbols it encodes
    q = oracle.quantise_ref(np.stack([oracle.normal_ref(oracle.stream_key_ref(9, c, 2), 20).reshape(4, 5) * (0.5 + c) for c in range(3)]), 0.25, 7)
    np.savez_compressed(os.path.join(HERE, "latent_stream.npz"), container=np.frombuffer(oracle.encode_latent_ref(q, 0.25, 7), np.uint8),
                        symbols=q.astype(np.int16), step=np.float32(0.25))
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
