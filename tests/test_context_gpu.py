"""Context front-end: latent z -> conditioning on the HIP path against the oracle restatement, and the
whole chain latent -> cond -> reverse diffusion against the oracle chain through both PSNR gates.  Needs a GPU: -m gpu."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def psnr(a, b):
    mse = ((a.double() - b.double()) ** 2).mean().item()
    return float("inf") if mse == 0 else 10.0 * math.log10(4.0 / mse)


@pytest.fixture(scope="module")
def cdx_mod(lib):
    import cdx
    assert torch.cuda.is_available()
    return cdx


CONCAT = dict(image_size=64, base_channels=32, channel_mult=(1, 2), attn_resolutions=(32,), num_res_blocks=1)
CROSS = dict(image_size=64, base_channels=64, channel_mult=(1, 2), cond_mode="cross_attn", attn_resolutions=(32,),
             cross_attn_resolutions=(64, 32), context_dim=96, num_res_blocks=1)


@pytest.mark.parametrize("name,over,ctx_over,split", [
    ("concat", CONCAT, dict(), True),
    ("concat_f32mfma", CONCAT, dict(), False),
    ("concat_wide_no_up", CONCAT, dict(hidden=128, num_blocks=3, upsample=0, latent_channels=32), True),
    ("cross", CROSS, dict(num_blocks=1), True),
    ("cfg2_shape", dict(), dict(), True),             # 256^2: latent 16 x 16 -> cond 64 x 64 (layers >= 16 px wide: split tile)
])
def test_context_forward_matches_oracle(cdx_mod, record, name, over, ctx_over, split):
    import oracle
    ucfg = cdx_mod.unet_config(**over)
    ccfg = cdx_mod.context_config(**ctx_over)
    params = cdx_mod.init_context_params(ucfg, ccfg, seed=3, affine_jitter=0.1)
    B = 2
    z = torch.from_numpy(cdx_mod.synthetic_latent(ccfg, ucfg["image_size"], 3, 0, B))
    want = oracle.context_forward_ref(ucfg, ccfg, params, z, dtype=torch.float64)
    ctx = cdx_mod.ContextNet(ucfg, ccfg, params, split=split)
    before = cdx_mod._abi.lib().cdx_launch_count()
    got = ctx(z.cuda()).cpu()
    # every recorded call launches (a convolution on an f32-MFMA tile that owes its consumer the per-image maxima launches twice:
    # the kernel and the cdx.h amax_out fallback pass)
    calls = ctx._plans[(B, z.shape[2], z.shape[3])].calls
    assert len(calls) <= cdx_mod._abi.lib().cdx_launch_count() - before <= len(calls) + len([c for c in calls if c[0].__name__ == "cdx_conv_f32"])
    assert got.shape == want.shape
    if ucfg["cond_mode"] == "cross_attn":
        assert got.shape == (B, (ucfg["image_size"] // 16) ** 2, ucfg["context_dim"])
    err, scale = (got.double() - want).abs().max().item(), want.abs().max().item()
    record("context_forward_" + name, hip_vs_fp64=err, scale=scale)
    assert err <= 2e-5 * max(1.0, scale), f"{name}: max err {err:.3e}"


@pytest.mark.parametrize("name,over,method", [("concat", CONCAT, "ddim"), ("cross", CROSS, "ddpm")])
def test_latent_to_image_chain_vs_oracle(cdx_mod, record, name, over, method):
    """latent -> context net -> sample(cond, steps): HIP chain vs oracle chain, both PSNR gates."""
    import oracle
    ucfg = cdx_mod.unet_config(**over)
    ccfg = cdx_mod.context_config(num_blocks=1)
    uparams = cdx_mod.init_params(ucfg, seed=9)
    cparams = cdx_mod.init_context_params(ucfg, ccfg, seed=9)
    B = 2
    z = torch.from_numpy(cdx_mod.synthetic_latent(ccfg, ucfg["image_size"], 9, 0, B))
    tgt = torch.from_numpy(cdx_mod.synthetic_batch(ucfg, 9, 0, B)["target"])
    sampler = cdx_mod.Sampler(cdx_mod.UNet(ucfg, uparams), method=method)
    got = cdx_mod.decode_latent(sampler, cdx_mod.ContextNet(ucfg, ccfg, cparams), z.cuda(), 6, seed=9).cpu()
    cond = oracle.context_forward_ref(ucfg, ccfg, cparams, z)
    want = oracle.sample_ref(ucfg, uparams, cond, 6, seed=9, method=method)
    record("latent_chain_" + name, psnr_hip_vs_oracle=psnr(got, want), dpsnr=abs(psnr(got, tgt) - psnr(want, tgt)))
    assert psnr(got, want) >= 80.0 and abs(psnr(got, tgt) - psnr(want, tgt)) <= 0.01
