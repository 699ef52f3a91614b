"""TEST INFRASTRUCTURE (PARITY UNPINNED: the reference snapshot holds no metric or image tooling to pin against).
CPU restatement of the image-side definitions behind cdx_export_u8 / cdx_psnr_f32 / cdx_msssim_f32 with stock torch:
8-bit quantisation, PSNR, and MS-SSIM (Wang, Simoncelli, Bovik 2003: 5 scales, 11 x 11 Gaussian window sigma 1.5 as a valid
F.conv2d, F.avg_pool2d(2) between scales, weights 0.0448 / 0.2856 / 0.3001 / 0.2363 / 0.1333).  Plus minimal PPM / PNG
readers for the round-trip tests of the product's writers."""
from __future__ import annotations

import struct
import zlib

import numpy as np
import torch
import torch.nn.functional as F

MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def to_uint8_ref(x_nchw: torch.Tensor, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    """[B,C,H,W] float -> uint8 [B,H,W,C]: floor((clamp(x) - lo) * 255 / (hi - lo) + 0.5), evaluated in float32 as the kernel does."""
    v = (x_nchw.float().clamp(lo, hi) - lo) * (255.0 / (hi - lo))
    return torch.floor(v + 0.5).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def psnr_ref(a: torch.Tensor, b: torch.Tensor, data_range: float = 2.0) -> torch.Tensor:
    d = (a.double() - b.double()).reshape(a.shape[0], -1)
    return 10.0 * torch.log10(data_range ** 2 / d.pow(2).mean(1))


def _gauss(dtype):
    k = torch.arange(11, dtype=torch.float64) - 5
    g = torch.exp(-k * k / (2 * 1.5 * 1.5))
    return (g / g.sum()).to(dtype)


def _ssim_maps(x, y, data_range):
    C = x.shape[1]
    g = _gauss(x.dtype)
    win = (g[:, None] * g[None, :]).expand(C, 1, 11, 11).contiguous()
    mu = lambda t: F.conv2d(t, win, groups=C)      # noqa: E731
    mx, my = mu(x), mu(y)
    vx, vy, cxy = mu(x * x) - mx * mx, mu(y * y) - my * my, mu(x * y) - mx * my
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    cs = (2 * cxy + c2) / (vx + vy + c2)
    return (2 * mx * my + c1) / (mx * mx + my * my + c1) * cs, cs


def msssim_ref(x: torch.Tensor, y: torch.Tensor, data_range: float = 2.0, dtype=torch.float64, return_scales: bool = False):
    x, y = x.to(dtype), y.to(dtype)
    vals = []
    for j in range(5):
        ssim, cs = _ssim_maps(x, y, data_range)
        vals.append((ssim if j == 4 else cs).mean(dim=(1, 2, 3)))
        if j < 4:
            x, y = F.avg_pool2d(x, 2), F.avg_pool2d(y, 2)
    v = torch.stack(vals, 1)
    w = torch.tensor(MS_WEIGHTS, dtype=dtype)
    out = (v.clamp_min(0) ** w).prod(1)
    return (out, v) if return_scales else out


def read_ppm_ref(data: bytes) -> np.ndarray:
    assert data[:2] in (b"P6", b"P5")
    parts, pos = [], 2
    while len(parts) < 3:
        while data[pos:pos + 1].isspace():
            pos += 1
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        parts.append(int(data[pos:end]))
        pos = end
    w, h, mx = parts
    assert mx == 255
    c = 3 if data[:2] == b"P6" else 1
    return np.frombuffer(data, np.uint8, h * w * c, pos + 1).reshape(h, w, c)


def read_png_ref(data: bytes) -> np.ndarray:
    """Decode the subset the product writes (8-bit grey / RGB, no interlace) -- every chunk's CRC is checked, all five PNG
    filter types are undone (a standard decoder's view of the file)."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack_from(">I4s", data, pos)
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack_from(">I", data, pos + 8 + n)[0] == (zlib.crc32(tag + body) & 0xFFFFFFFF), tag
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype, comp, flt, inter = hdr
    assert depth == 8 and ctype in (0, 2) and comp == 0 and flt == 0 and inter == 0
    c = 3 if ctype == 2 else 1
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * c)
    out = np.zeros((h, w * c), np.int64)
    for r in range(h):
        f, line = int(raw[r, 0]), raw[r, 1:].astype(np.int64)
        prev = out[r - 1] if r else np.zeros(w * c, np.int64)
        if f == 0:
            out[r] = line
        elif f == 2:
            out[r] = (line + prev) & 255
        else:
            for i in range(w * c):
                a = out[r, i - c] if i >= c else 0
                b, cc = prev[i], (prev[i - c] if i >= c else 0)
                if f == 1:
                    pred = a
                elif f == 3:
                    pred = (a + b) // 2
                else:
                    p = a + b - cc
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - cc)
                    pred = a if pa <= pb and pa <= pc else b if pb <= pc else cc
                out[r, i] = (line[i] + pred) & 255
    return out.astype(np.uint8).reshape(h, w, c)
