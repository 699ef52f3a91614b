"""(f4) Bitstream side on the GPU: the rANS decode kernel (csrc/rans.hip) against the oracle decoder bit for bit, the
committed golden container, corrupt payloads, and the whole chain bytes -> latent -> cond -> image against the oracle chain.
Needs a GPU: run with -m gpu."""
import math
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "latent_stream.npz")


def psnr(a, b):
    mse = ((a.double() - b.double()) ** 2).mean().item()
    return float("inf") if mse == 0 else 10.0 * math.log10(4.0 / mse)


@pytest.fixture(scope="module")
def cdx_mod(lib):
    import cdx
    assert torch.cuda.is_available()
    return cdx


def test_golden_container_on_gpu(cdx_mod):
    g = np.load(GOLD)
    z, sym = cdx_mod.LatentDecoder()([g["container"].tobytes()], return_symbols=True)
    assert tuple(z.shape) == (1, 3, 4, 5)
    assert np.array_equal(sym.cpu().numpy()[0], g["symbols"])
    assert np.array_equal(z.cpu().numpy()[0], g["symbols"].astype(np.float32) * g["step"])


@pytest.mark.parametrize("B,cz,h,w,qmax,pb,scale", [(3, 16, 16, 16, 15, 12, 1.0), (1, 1, 1, 1, 3, 8, 1.0), (2, 3, 16, 16, 127, 12, 25.0),
                                                    (2, 2, 7, 3, 0, 4, 1.0), (5, 5, 8, 8, 31, 10, 0.05), (70, 4, 8, 8, 15, 12, 2.0),
                                                    (1, 16, 32, 32, 15, 12, 0.6)])
def test_rans_kernel_matches_oracle_bit_for_bit(cdx_mod, B, cz, h, w, qmax, pb, scale):
    rng = np.random.default_rng(B * 1000 + cz)
    qs = oracle.quantise_ref(rng.standard_normal((B, cz, h, w)) * scale, 0.5, qmax)
    freq = oracle.build_freq_ref(qs, qmax, pb)                                   # one table for the batch
    bufs = [oracle.encode_latent_ref(qs[b], 0.5, qmax, pb, freq) for b in range(B)]
    before = cdx_mod._abi.lib().cdx_launch_count()
    z, sym = cdx_mod.LatentDecoder()(bufs, return_symbols=True)
    assert cdx_mod._abi.lib().cdx_launch_count() == before + 1
    want = np.stack([oracle.decode_latent_ref(b)[0] for b in bufs])
    assert np.array_equal(want, qs)
    assert np.array_equal(sym.cpu().numpy(), want) and np.array_equal(z.cpu().numpy(), want.astype(np.float32) * np.float32(0.5))


def test_corrupt_payload_is_reported_not_crashed(cdx_mod):
    g = bytearray(np.load(GOLD)["container"].tobytes())
    for flip in (len(g) - 1, len(g) - 7, len(g) - 20):
        bad = bytearray(g)
        bad[flip] ^= 0x5A
        try:
            z = cdx_mod.LatentDecoder()([bytes(bad)])
        except ValueError as e:
            assert "corrupt" in str(e)
        else:                                               # a flipped payload bit may still decode to SOME symbols of a valid
            assert torch.isfinite(z).all()                  # stream only if the final-state check passes by chance (2^-16)
    with pytest.raises(ValueError, match="share"):
        other = oracle.encode_latent_ref(np.zeros((3, 4, 5), np.int64), 0.5, 7)
        cdx_mod.LatentDecoder()([bytes(g), other])


def test_device_refuses_a_bad_frequency_table_and_stops_on_a_short_stream(cdx_mod):
    """the kernel trusts no table (sum != 2^prob_bits -> status, nothing decoded: no slot without a symbol is ever
    looked up) and a stream that runs dry stops instead of spinning through its remaining symbols."""
    A = cdx_mod._abi
    rng = np.random.default_rng(5)
    q = oracle.quantise_ref(rng.standard_normal((2, 16, 16)) * 3.0, 0.5, 15)
    g = cdx_mod.parse_latent_stream(oracle.encode_latent_ref(q, 0.5, 15, 12))
    dev = "cuda"
    up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)      # noqa: E731
    nsym, cz = g["height"] * g["width"], g["channels"]
    words, off = up(g["words"].view(np.int16), np.int16), up(g["off"], np.int32)
    for case in ("good", "bad_table", "short_stream"):
        freq = g["freq"].copy()
        lens = g["len"].copy()
        if case == "bad_table":
            freq[0] -= 1                                        # sums to 2^pb - 1: slot 2^pb - 1 has no symbol
        elif case == "short_stream":
            lens[1] = 2                                         # stream 1 holds only its initial state: it runs dry within ~10 symbols
        out = torch.full((cz, nsym), 7.0, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        d_len, d_freq = up(lens, np.int32), up(freq.view(np.int16), np.int16)      # (held: a temporary's block would be reused at once)
        a = A.RansDecodeArgs(words.data_ptr(), off.data_ptr(), d_len.data_ptr(), d_freq.data_ptr(),
                             cz, nsym, 2 * g["qmax"] + 1, g["prob_bits"], g["qmax"], g["step"], out.data_ptr(), None, status.data_ptr())
        A.call("rans_decode_i16", a, None, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want = torch.from_numpy(q.reshape(cz, nsym).astype(np.float32) * np.float32(0.5)).to(dev)
        assert int(status.item()) == (0 if case == "good" else 1), case
        assert torch.isfinite(out).all()
        if case == "good":
            assert torch.equal(out, want)
        elif case == "bad_table":
            assert (out == 0).all()                             # nothing decoded
        else:
            assert torch.equal(out[0], want[0])                 # the intact stream is decoded in full
            assert (out[1, 64:] == 0).all() and torch.equal(out[1, :2], want[1, :2])      # the short one stopped early, zero-filled
    with pytest.raises(ValueError, match="no containers"):
        cdx_mod.LatentDecoder()([])


def test_bytes_to_image_chain_vs_oracle(cdx_mod, record):
    """bitstream -> rANS decode -> context net -> 6 DDIM steps, HIP vs the oracle chain: both PSNR gates."""
    ucfg = cdx_mod.unet_config(image_size=64, base_channels=32, channel_mult=(1, 2), attn_resolutions=(32,), num_res_blocks=1)
    ccfg = cdx_mod.context_config(num_blocks=1)
    uparams, cparams = cdx_mod.init_params(ucfg, seed=9), cdx_mod.init_context_params(ucfg, ccfg, seed=9)
    B = 2
    zsrc = cdx_mod.synthetic_latent(ccfg, ucfg["image_size"], 9, 0, B)
    qs = oracle.quantise_ref(zsrc, 0.125, 31)
    freq = oracle.build_freq_ref(qs, 31, 12)
    bufs = [oracle.encode_latent_ref(qs[b], 0.125, 31, 12, freq) for b in range(B)]
    sampler = cdx_mod.Sampler(cdx_mod.UNet(ucfg, uparams))
    got = cdx_mod.decode_bitstreams(sampler, cdx_mod.ContextNet(ucfg, ccfg, cparams), bufs, 6, seed=9).cpu()
    zq = torch.from_numpy(np.stack([oracle.decode_latent_ref(b)[1] for b in bufs]))
    want = oracle.sample_ref(ucfg, uparams, oracle.context_forward_ref(ucfg, ccfg, cparams, zq), 6, seed=9)
    tgt = torch.from_numpy(cdx_mod.synthetic_batch(ucfg, 9, 0, B)["target"])
    record("bytes_to_image_chain", psnr_hip_vs_oracle=psnr(got, want), dpsnr=abs(psnr(got, tgt) - psnr(want, tgt)),
           bytes_per_image=len(bufs[0]))
    assert psnr(got, want) >= 80.0 and abs(psnr(got, tgt) - psnr(want, tgt)) <= 0.01
