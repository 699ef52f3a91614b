"""Host-side contract (config, graph, schedule, generator, synthetic data): CPU only."""
import math

import numpy as np
import pytest

import cdx


def test_named_configs_match_baseline_shapes():
    c1, r1 = cdx.named_config("cfg1")
    assert (c1["image_size"], c1["base_channels"], r1) == (32, 64, dict(batch=1, steps=50, method="ddim"))
    c2, r2 = cdx.named_config("cfg2")
    assert (c2["image_size"], c2["base_channels"], c2["attn_resolutions"], r2["batch"], r2["steps"]) == (256, 128, (16,), 16, 100)
    assert cdx.named_config("cfg3")[1]["batch"] == 128
    c4, r4 = cdx.named_config("cfg4")
    assert (c4["image_size"], c4["base_channels"], c4["cond_mode"], r4) == (512, 192, "cross_attn", dict(batch=8, steps=250, method="ddpm"))


def test_parameter_counts_match_survey():
    # 9.0 M / 113.7 M / 274.1 M parameters
    for name, want in (("cfg1", 8953795), ("cfg2", 113676675), ("cfg4", 274111491)):
        g = cdx.build_graph(cdx.named_config(name)[0])
        assert sum(int(np.prod(s)) for s in g.param_shapes.values()) == want


def test_graph_block_inventory_cfg2():
    g = cdx.build_graph(cdx.named_config("cfg2")[0])
    blocks = g.down + g.mid + g.up
    kinds = [b.kind for b in blocks]
    assert kinds.count("res") == 32 and kinds.count("attn") == 6 and kinds.count("down") == 5 and kinds.count("up") == 5
    assert sum(1 for b in blocks if b.kind == "res" and b.cin != b.cout) == 20       # 1x1 skip convs
    assert all(b.skip_ch > 0 for b in g.up if b.kind == "res")


def test_config_validation():
    with pytest.raises(KeyError):
        cdx.unet_config(bogus=1)
    with pytest.raises(ValueError):
        cdx.unet_config(cond_mode="film")
    with pytest.raises(ValueError):
        cdx.unet_config(image_size=100)           # not divisible by 2**5
    with pytest.raises(ValueError):
        cdx.unet_config(base_channels=48)         # 48 % 32 groups
    with pytest.raises(ValueError):
        cdx.unet_config(dtype="fp8")
    assert cdx.unet_config(dtype="fp16")["dtype"] == "fp16"
    c5, r5 = cdx.named_config("cfg5")
    assert c5["dtype"] == "fp16" and c5["image_size"] == 256 and r5["image"] == 1024 and r5["steps"] == 50


def test_schedule_identities():
    s = cdx.make_schedule()
    ab, b = s["alphas_cumprod"], s["betas"]
    assert len(ab) == 1000 and np.all(np.diff(ab) < 0)
    assert abs(ab[0] - (1 - 1e-4)) < 1e-15 and b[0] == 1e-4 and abs(b[-1] - 2e-2) < 1e-15
    assert 3e-5 < ab[-1] < 6e-5
    assert list(cdx.timestep_subsequence(1000, 100)[:3]) == [0, 10, 20]
    assert list(cdx.timestep_subsequence(1000, 1000)) == list(range(1000))
    with pytest.raises(ValueError):
        cdx.timestep_subsequence(1000, 0)


def test_ddim_step_algebra_float64():
    """x_prev = sqrt(ab_p) x0h + sqrt(1-ab_p) eps and x0h = (x - sqrt(1-ab) eps)/sqrt(ab), in closed form."""
    s = cdx.make_schedule()
    ab = s["alphas_cumprod"]
    coefs = cdx.step_coefficients(s, 50, "ddim")
    assert [c.t for c in coefs] == list(range(980, -1, -20))
    rng = np.random.default_rng(0)
    x, e = rng.standard_normal(64), rng.standard_normal(64)
    for k in (0, 20, 49):
        c = coefs[k]
        x0 = (x - math.sqrt(1 - ab[c.t]) * e) / math.sqrt(ab[c.t])
        abp = ab[c.t - 20] if c.t >= 20 else 1.0
        want = math.sqrt(abp) * x0 + math.sqrt(1 - abp) * e
        got = c.cx * x + c.c0 * (c.ca * x + c.cb * e) + c.ce * e
        assert np.allclose(got, want, rtol=1e-12, atol=1e-12)
    assert coefs[-1].c0 == 1.0 and coefs[-1].ce == 0.0            # last step lands on x0h exactly


def test_ddpm_coefficients_match_epsilon_form():
    """Unclipped posterior mean equals (x - b/sqrt(1-ab) eps)/sqrt(a) (Ho et al. eq. 11), sigma^2 = beta-tilde."""
    s = cdx.make_schedule()
    ab = s["alphas_cumprod"]
    coefs = cdx.step_coefficients(s, 1000, "ddpm")
    rng = np.random.default_rng(1)
    x, e = rng.standard_normal(16), rng.standard_normal(16)
    for k in (0, 500, 998):
        c = coefs[k]
        a_t = ab[c.t] / ab[c.t - 1]
        want = (x - (1 - a_t) / math.sqrt(1 - ab[c.t]) * e) / math.sqrt(a_t)
        got = c.cx * x + c.c0 * (c.ca * x + c.cb * e)
        assert np.allclose(got, want, rtol=1e-9)
        assert abs(c.sigma ** 2 - (1 - a_t) * (1 - ab[c.t - 1]) / (1 - ab[c.t])) < 1e-15
    assert coefs[-1].sigma == 0.0


def test_generator_is_deterministic_and_normal():
    k = cdx.rng.stream_key(0, 0, 1)
    a, b = cdx.rng.normal(k, 100000), cdx.rng.normal(k, 100000)
    assert a.dtype == np.float32 and np.array_equal(a, b)
    assert abs(a.mean()) < 0.01 and abs(a.std() - 1) < 0.01
    assert np.array_equal(cdx.rng.normal(k, 10, offset=5), a[5:15])
    assert cdx.rng.stream_key(0, 0, 1) != cdx.rng.stream_key(0, 1, 1) != cdx.rng.stream_key(1, 0, 1)
    u = cdx.rng.uniform(k, 100000)
    assert 0 < u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.005


def test_init_params_deterministic_and_scaled():
    cfg = cdx.unet_config(image_size=16, base_channels=32, channel_mult=(1, 2), attn_resolutions=(8,), num_res_blocks=1)
    p0, p1, p2 = cdx.init_params(cfg, 3), cdx.init_params(cfg, 3), cdx.init_params(cfg, 4)
    assert all(np.array_equal(p0[k], p1[k]) for k in p0)
    assert not np.array_equal(p0["conv_in.weight"], p2["conv_in.weight"])
    w = p0["down.0.0.res.conv1.weight"]
    assert np.abs(w).max() <= 1 / math.sqrt(32 * 9) + 1e-7
    assert np.all(p0["out.norm.weight"] == 1) and np.all(p0["out.norm.bias"] == 0)
    full = cdx.init_params(cfg, 3, out_gain=1.0)
    assert np.allclose(p0["out.conv.weight"], full["out.conv.weight"] * cdx.params.OUT_GAIN, rtol=1e-6)


def test_synthetic_batch_keyed_by_global_index():
    cfg = cdx.unet_config(image_size=32, base_channels=64, channel_mult=(1, 2, 2, 2))
    whole, part = cdx.synthetic_batch(cfg, 9, 0, 4), cdx.synthetic_batch(cfg, 9, 2, 2)
    assert np.array_equal(whole["cond"][2:], part["cond"]) and np.array_equal(whole["target"][2:], part["target"])
    assert whole["cond"].shape == (4, 3, 2, 2) and np.abs(whole["target"]).max() <= 1


def test_tile_plan_matches_survey_and_oracle():
    import oracle
    ys, xs = cdx.tile_plan(1024, 1024, 256, 64)
    assert ys == xs == [0, 192, 384, 576, 768] and len(ys) * len(xs) == 25          # 25 tiles with 64-px overlap
    assert cdx.tile_plan(1024, 1024, 256, 0)[0] == [0, 256, 512, 768]                # 16 tiles without overlap
    assert cdx.tile_origins(80, 32, 16) == [0, 16, 32, 48] == oracle.origins_ref(80, 32, 16)
    assert cdx.tile_origins(256, 256, 64) == [0]
    with pytest.raises(ValueError):
        cdx.tile_origins(80, 32, 8)        # origins 24, 48 are not multiples of the conditioning stride
    with pytest.raises(ValueError):
        cdx.tile_origins(16, 32, 0)
