"""The CPU oracle against the committed golden fixtures and its own invariants.  CPU only.

PARITY UNPINNED by the reference (it ships no code or vectors): these fixtures were produced by the oracle
itself (tests/golden/make_golden.py) and pin the build's contract across rounds and library versions."""
import math
import os

import numpy as np
import pytest
import torch

import cdx
import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TINY = dict(image_size=16, base_channels=32, channel_mult=(1, 2), attn_resolutions=(8,), num_res_blocks=1)


def psnr(a, b):
    mse = ((a.double() - b.double()) ** 2).mean().item()
    return float("inf") if mse == 0 else 10.0 * math.log10(4.0 / mse)


def test_generator_golden_and_product_agree():
    g = np.load(os.path.join(GOLD, "rng.npz"))
    triples = [(0, 0, 1), (7, 3, 16), (123456789, 1 << 40, 0)]
    for i, (s, a, b) in enumerate(triples):
        k = oracle.stream_key_ref(s, a, b)
        assert k == int(g["keys"][i]) == cdx.rng.stream_key(s, a, b)
        assert np.array_equal(oracle.normal_ref(k, 1024), g["normal"][i])
        assert np.array_equal(cdx.rng.normal(k, 1024), g["normal"][i])
        assert np.array_equal(oracle.uniform_ref(k, 1024), g["uniform"][i])
        assert np.array_equal(cdx.rng.uniform(k, 1024), g["uniform"][i])
    # scalar (pure Python integer) restatement
    from oracle.rng_ref import normal_scalar
    k = oracle.stream_key_ref(0, 0, 1)
    assert [normal_scalar(k, i) for i in range(8)] == [float(v) for v in g["normal"][0][:8]]


def test_schedule_golden_and_product_agree():
    g = np.load(os.path.join(GOLD, "schedule.npz"))
    s = cdx.make_schedule()
    assert np.allclose(s["alphas_cumprod"], g["alphas_cumprod"], rtol=1e-13)
    for name, steps, method in (("ddim100", 100, "ddim"), ("ddpm250", 250, "ddpm")):
        want = g[name]
        got = np.array([(c.t, c.ca, c.cb, c.cx, c.c0, c.ce, c.sigma) for c in cdx.step_coefficients(s, steps, method)])
        ref = np.array(oracle.step_coefficients_ref(steps, method))
        assert np.allclose(ref, want, rtol=1e-13, atol=0) and np.allclose(got, want, rtol=1e-12, atol=1e-15)


def test_timestep_embedding_definition():
    t = torch.tensor([0, 1, 999])
    e = oracle.timestep_embedding_ref(t, 128, torch.float64)
    assert e.shape == (3, 128)
    assert torch.allclose(e[0, :64], torch.zeros(64, dtype=torch.float64)) and torch.allclose(e[0, 64:], torch.ones(64, dtype=torch.float64))
    assert abs(e[1, 0].item() - math.sin(1.0)) < 1e-15 and abs(e[1, 63].item() - math.sin(1e-4)) < 1e-15


def test_oracle_tiny_ddpm_golden():
    """8-step ancestral sampling on the tiny net reproduces the committed vector (noise floor ~1e-6)."""
    g = np.load(os.path.join(GOLD, "tiny_ddpm.npz"))
    cfg = cdx.unet_config(**TINY)
    params = cdx.init_params(cfg, seed=5, affine_jitter=0.1)
    cond = torch.from_numpy(cdx.synthetic_batch(cfg, 5, 0, 2)["cond"])
    x0 = oracle.sample_ref(cfg, params, cond, 8, seed=5, method="ddpm")
    assert psnr(x0, torch.from_numpy(g["x0"])) > 100.0


def test_oracle_cfg1_first_step_golden():
    """BASELINE.json configs[0] (32x32, 64-ch, batch 1): the first UNet call and the first DDIM step."""
    g = np.load(os.path.join(GOLD, "cfg1_ddim50.npz"))
    cfg, run = cdx.named_config("cfg1")
    params = cdx.init_params(cfg, seed=0)
    cond = torch.from_numpy(cdx.synthetic_batch(cfg, 0, 0, 1)["cond"])
    xT = oracle.sampler_ref.noise_ref(0, 0, 1, 1, (3, 32, 32))
    eps = oracle.unet_forward_ref(cfg, params, xT, torch.full((1,), int(g["t0"]), dtype=torch.int64), cond)
    assert (eps - torch.from_numpy(g["eps0"])).abs().max().item() < 5e-6
    trace = []
    oracle.sample_ref(cfg, params, cond, 1, seed=0, trace=trace)     # a 1-step chain: different tau, only runs the code
    assert trace[0].shape == (1, 3, 32, 32)


@pytest.mark.timeout(300)
def test_oracle_cfg1_full_golden_and_noise_floor():
    """Full 50-step decode: reproduces the fixture, and the single-thread run agrees with it to > 100 dB --
    the noise floor that makes the 80 dB / 0.01 dB gates of the GPU tests meaningful."""
    g = np.load(os.path.join(GOLD, "cfg1_ddim50.npz"))
    cfg, run = cdx.named_config("cfg1")
    params = cdx.init_params(cfg, seed=0)
    sb = cdx.synthetic_batch(cfg, 0, 0, 1)
    cond, tgt = torch.from_numpy(sb["cond"]), torch.from_numpy(sb["target"])
    n = torch.get_num_threads()
    try:
        torch.set_num_threads(1)
        x1 = oracle.sample_ref(cfg, params, cond, run["steps"], seed=0)
    finally:
        torch.set_num_threads(n)
    want = torch.from_numpy(g["x0"])
    assert psnr(x1, want) > 100.0
    assert abs(psnr(x1, tgt) - psnr(want, tgt)) < 1e-3
    assert x1.abs().max().item() <= 1.0


def test_oracle_determinism_and_fp64_agreement():
    cfg = cdx.unet_config(**TINY)
    params = cdx.init_params(cfg, seed=1)
    cond = torch.from_numpy(cdx.synthetic_batch(cfg, 1, 0, 1)["cond"])
    a = oracle.sample_ref(cfg, params, cond, 5, seed=1)
    b = oracle.sample_ref(cfg, params, cond, 5, seed=1)
    c = oracle.sample_ref(cfg, params, cond, 5, seed=1, dtype=torch.float64)
    assert torch.equal(a, b) and psnr(a, c) > 110.0


def test_oracle_cross_attention_runs():
    cfg = cdx.unet_config(image_size=32, base_channels=64, channel_mult=(1, 2), cond_mode="cross_attn",
                          attn_resolutions=(16,), cross_attn_resolutions=(32, 16), context_dim=64, num_res_blocks=1)
    params = cdx.init_params(cfg, seed=2)
    cond = torch.from_numpy(cdx.synthetic_batch(cfg, 2, 0, 1)["cond"])
    assert cond.shape == (1, 4, 64)
    x0 = oracle.sample_ref(cfg, params, cond, 2, seed=2, method="ddpm")
    assert x0.shape == (1, 3, 32, 32) and torch.isfinite(x0).all()
