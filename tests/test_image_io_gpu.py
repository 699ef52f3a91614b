"""(f4) Image side on the GPU: cdx_export_u8, cdx_psnr_f32 and cdx_msssim_f32 against the oracle's stock-torch restatement
(oracle/metrics_ref.py), and decode_bitstreams(..., out_path=) end to end: bytes -> latent -> cond -> image -> PNG on disk whose
pixels are the 8-bit quantisation of the returned tensor.  Needs a GPU: -m gpu."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cdx_mod(lib):
    import cdx
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return cdx


def test_export_u8_is_bit_exact(cdx_mod):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 3, 37, 53, generator=g) * 0.8
    x[0, 0, 0, :7] = torch.tensor([-1.0, 1.0, 0.0, -2.0, 3.0, -1 + 2 / 255 * 0.5, float("nan")])
    want = oracle.to_uint8_ref(torch.nan_to_num(x, nan=-1.0))              # NaN -> lo, as the kernel documents
    got = cdx_mod.to_uint8(x.cuda()).cpu()
    assert got.dtype == torch.uint8 and torch.equal(got, want)
    # straight from an NHWC state buffer with padding channels (the sampler's layout): no layout copy
    buf = torch.full((3, 37, 53, 8), 9.0, device="cuda")
    buf[..., :3] = x.permute(0, 2, 3, 1)
    assert torch.equal(cdx_mod.to_uint8(buf, nhwc_channels=3).cpu(), want)


@pytest.mark.parametrize("shape", [(2, 3, 256, 256), (1, 3, 176, 300), (3, 1, 200, 177), (1, 3, 512, 384)], ids=lambda s: "x".join(map(str, s)))
def test_psnr_and_msssim_match_the_oracle(cdx_mod, record, shape):
    g = torch.Generator().manual_seed(sum(shape))
    B = shape[0]
    # smooth "images" + noise of a different level per image
    base = torch.nn.functional.interpolate(torch.rand(B, shape[1], 16, 16, generator=g), size=shape[2:], mode="bicubic", align_corners=False).clamp(0, 1) * 2 - 1
    noise = torch.randn(shape, generator=g) * torch.tensor([0.02, 0.1, 0.3][:B]).reshape(B, 1, 1, 1)
    y = (base + noise).clamp(-1, 1)
    p = cdx_mod.psnr(base.cuda(), y.cuda()).cpu().double()
    assert torch.allclose(p, oracle.psnr_ref(base, y), rtol=0, atol=2e-5)
    m, scales = cdx_mod.ms_ssim(base.cuda(), y.cuda(), return_scales=True)
    want, wscales = oracle.msssim_ref(base, y, return_scales=True)
    err, serr = (m.cpu().double() - want).abs().max().item(), (scales.cpu().double() - wscales).abs().max().item()
    e32 = (oracle.msssim_ref(base, y, dtype=torch.float32).double() - want).abs().max().item()
    record("msssim", shape="x".join(map(str, shape)), err=err, per_scale_err=serr, cpu_fp32_err=e32, value=want.tolist())
    assert err <= 2e-5 and serr <= 2e-5
    assert torch.equal(cdx_mod.ms_ssim(base.cuda(), base.cuda()).cpu(), torch.ones(B))
    assert cdx_mod.psnr(base.cuda(), base.cuda()).isinf().all()
    # deterministic: the same bits from launch to launch (fixed summation order, no atomics)
    assert torch.equal(cdx_mod.ms_ssim(base.cuda(), y.cuda()), m)


def test_msssim_rejects_images_too_small_for_five_scales(cdx_mod):
    x = torch.zeros(1, 3, 128, 256, device="cuda")
    with pytest.raises(ValueError, match="176"):
        cdx_mod.ms_ssim(x, x)


def test_decode_bitstreams_writes_the_image_it_returns(cdx_mod, tmp_path):
    """bytes -> rANS decode -> context net -> 4 DDIM steps -> PNG / PPM files: the files hold exactly the 8-bit quantisation (oracle
    definition) of the float tensor the call returns."""
    ucfg = cdx_mod.unet_config(image_size=64, base_channels=32, channel_mult=(1, 2), attn_resolutions=(32,), num_res_blocks=1)
    ccfg = cdx_mod.context_config(num_blocks=1)
    uparams, cparams = cdx_mod.init_params(ucfg, seed=9), cdx_mod.init_context_params(ucfg, ccfg, seed=9)
    B = 2
    qs = oracle.quantise_ref(cdx_mod.synthetic_latent(ccfg, ucfg["image_size"], 9, 0, B), 0.125, 31)
    freq = oracle.build_freq_ref(qs, 31, 12)
    bufs = [oracle.encode_latent_ref(qs[b], 0.125, 31, 12, freq) for b in range(B)]
    sampler, ctx = cdx_mod.Sampler(cdx_mod.UNet(ucfg, uparams)), cdx_mod.ContextNet(ucfg, ccfg, cparams)
    for ext, rd in ((".png", oracle.read_png_ref), (".ppm", oracle.read_ppm_ref)):
        out = cdx_mod.decode_bitstreams(sampler, ctx, bufs, 4, seed=9, out_path=str(tmp_path / f"dec{ext}")).cpu()
        assert torch.equal(out, cdx_mod.decode_bitstreams(sampler, ctx, bufs, 4, seed=9).cpu())      # same tensor with or without the files
        want = oracle.to_uint8_ref(out)
        for b in range(B):
            img = rd((tmp_path / f"dec_{b:04d}{ext}").read_bytes())
            assert img.shape == (64, 64, 3) and np.array_equal(img, want[b].numpy())
