"""(f4) Bitstream side of the decode path: container parsing on the host and the rANS entropy
decode of the quantised latent on the GPU (csrc/rans.hip) -- the step before the context front-end:

    bytes  --parse-->  streams  --cdx_rans_decode_i16-->  z [B, Cz, h, w]  --ContextNet-->  cond  --Sampler.sample-->  x_0

Build-defined format (the reference snapshot holds no bitstream format: README.md is 0 bytes), "CDXL" version 1, one
container per image, little endian:

    0   4s   magic "CDXL"
    4   u16  version (1)          6   u16  prob_bits (<= 12)
    8   u16  channels Cz         10   u16  height h      12  u16  width w      14  u16  qmax
    16  f32  step (dequantisation step: z = symbol * step, symbol in [-qmax, qmax])
    20  u32  payload words (16-bit)
    24  u16  freq[2 qmax + 1] (sum = 2^prob_bits, every entry >= 1), padded with one zero u16 to a 4-byte boundary
    ..  u32  stream_off[Cz]   (first word of channel c's stream, relative to the payload)
    ..  u32  stream_len[Cz]
    ..  u16  payload[...]     each stream: final rANS state (high word, low word), then the words in decode order

There is no encoder in the product (decode path only); the oracle holds one for tests (oracle/bitstream_ref.py).
"""
from __future__ import annotations

import struct

import numpy as np
import torch

from . import _abi

MAGIC, VERSION = b"CDXL", 1


MAX_LATENT_SIDE, MAX_LATENT_CHANNELS = 1024, 1024      # reader limits (16384^2 images)


def parse_latent_stream(buf: bytes) -> dict:
    """Validate and split one CDXL container (host, numpy views; raises ValueError on any inconsistency)."""
    buf = bytes(buf)
    if len(buf) < 24 or buf[:4] != MAGIC:
        raise ValueError("not a CDXL latent stream (bad magic)")
    version, pb, cz, h, w, qmax = struct.unpack_from("<6H", buf, 4)
    step, nwords = struct.unpack_from("<fI", buf, 16)
    if version != VERSION:
        raise ValueError(f"CDXL version {version} is not supported (this reader: {VERSION})")
    alphabet = 2 * qmax + 1
    if not (1 <= pb <= 12) or alphabet > (1 << pb) or cz == 0 or h == 0 or w == 0 or not np.isfinite(step):
        raise ValueError("CDXL header out of range")
    # an untrusted 40-byte container must not be able to request gigabytes or a minutes-long serial decode:
    # latents are images / 16 (8192^2 images: 512 x 512), and a symbol costs at least ~1/16 bit of payload per the
    # table's most probable symbol (freq <= 2^pb - alphabet + 1), so nsym is bounded by the payload it is decoded from
    if h > MAX_LATENT_SIDE or w > MAX_LATENT_SIDE or cz > MAX_LATENT_CHANNELS:
        raise ValueError(f"CDXL latent {cz} x {h} x {w} exceeds the reader's limits ({MAX_LATENT_CHANNELS} x {MAX_LATENT_SIDE}^2)")
    pos = 24
    nfreq = alphabet + (alphabet & 1)
    need = pos + 2 * nfreq + 8 * cz + 2 * nwords
    if len(buf) != need:
        raise ValueError(f"CDXL container is {len(buf)} bytes, header implies {need}")
    freq = np.frombuffer(buf, "<u2", alphabet, pos)
    pos += 2 * nfreq
    off = np.frombuffer(buf, "<u4", cz, pos)
    pos += 4 * cz
    ln = np.frombuffer(buf, "<u4", cz, pos)
    pos += 4 * cz
    words = np.frombuffer(buf, "<u2", nwords, pos)
    if int(freq.astype(np.int64).sum()) != (1 << pb) or (freq == 0).any():
        raise ValueError("CDXL frequency table does not sum to 2^prob_bits or has an empty symbol")
    if ((off.astype(np.int64) + ln) > nwords).any() or (ln < 2).any():
        raise ValueError("CDXL stream table points outside the payload")
    # each renormalisation word carries 16 bits; a symbol of probability p costs -log2 p bits: h*w symbols of the MOST probable
    # symbol need at least h*w * -log2(fmax / 2^pb) bits (minus the 32-bit final state)
    fmax = int(freq.max())
    if fmax < (1 << pb):
        min_bits = h * w * -np.log2(fmax / float(1 << pb))
        if (16.0 * ln.astype(np.float64) + 32.0 < min_bits - 64.0).any():
            raise ValueError("CDXL stream is too short for the symbol count its header claims")
    return dict(prob_bits=pb, channels=cz, height=h, width=w, qmax=qmax, step=float(step), freq=freq, off=off, len=ln, words=words)


class LatentDecoder:
    """z = LatentDecoder(device)(list of CDXL containers)  ->  float32 [B, Cz, h, w] on the device (+ int16 symbols)."""

    def __init__(self, device="cuda"):
        _abi.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("LatentDecoder (HIP backend) needs a GPU; there is no CPU fallback in the product path")
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())

    @torch.no_grad()
    def decode(self, containers, return_symbols: bool = False):
        parsed = [parse_latent_stream(b) for b in containers]
        if not parsed:
            raise ValueError("LatentDecoder.decode: no containers")
        p0 = parsed[0]
        for p in parsed[1:]:
            same = all(p[k] == p0[k] for k in ("prob_bits", "channels", "height", "width", "qmax", "step")) and \
                np.array_equal(p["freq"], p0["freq"])
            if not same:
                raise ValueError("the containers of one batch must share geometry, step and frequency table")
        B, cz, h, w = len(parsed), p0["channels"], p0["height"], p0["width"]
        base, offs, lens, words = 0, [], [], []
        for p in parsed:
            offs.append(p["off"].astype(np.int64) + base)
            lens.append(p["len"])
            words.append(p["words"])
            base += p["words"].size
        dev = self.device
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)      # noqa: E731
        d_words = up(np.concatenate(words).view(np.int16), np.int16)
        d_off, d_len = up(np.concatenate(offs), np.int32), up(np.concatenate(lens), np.int32)      # (< 2^31 words)
        d_freq = up(p0["freq"].view(np.int16), np.int16)
        with torch.cuda.device(dev):
            out = torch.empty(B * cz, h * w, device=dev, dtype=torch.float32)
            sym = torch.empty(B * cz, h * w, device=dev, dtype=torch.int16) if return_symbols else None
            status = torch.zeros(1, device=dev, dtype=torch.int32)
            a = _abi.RansDecodeArgs(d_words.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), d_freq.data_ptr(), B * cz, h * w,
                                    2 * p0["qmax"] + 1, p0["prob_bits"], p0["qmax"], p0["step"], out.data_ptr(),
                                    sym.data_ptr() if sym is not None else None, status.data_ptr())
            _abi.call("rans_decode_i16", a, None, 0, torch.cuda.current_stream(dev).cuda_stream)
            if int(status.item()):
                raise ValueError("corrupt CDXL payload: a stream ran past its length or did not end in the initial rANS state")
        z = out.view(B, cz, h, w)
        return (z, sym.view(B, cz, h, w)) if return_symbols else z

    __call__ = decode


@torch.no_grad()
def decode_bitstreams(sampler, ctx, containers, steps: int, *, out_path: str | None = None, **kw) -> torch.Tensor:
    """bytes -> latent -> conditioning -> reverse diffusion: the whole decode path of the codec's synthesis side.
    out_path ("x.png" / "x.ppm"): also write the decoded images as 8-bit files (one: the path as given; several:
    x_0000.png, ...), quantised on the device straight from the sampler's state buffer (cdx_export_u8)."""
    z = LatentDecoder(sampler.unet.device)(containers)
    cond = ctx(z)
    if out_path is None:
        return sampler.sample(cond, steps, **kw)
    from .image_io import to_uint8, write_images
    # (the ONE sampling loop is Sampler.sample: the files are written from its state buffer through the on_finish hook)
    return sampler.sample(cond, steps, on_finish=lambda run: write_images(out_path, to_uint8(run.plan.xin, nhwc_channels=run.channels)), **kw)
