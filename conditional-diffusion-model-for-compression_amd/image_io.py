"""(f4) Image side of the decode path: 8-bit export on the device, dependency-free PPM / PNG
writers on the host, and the two quality metrics (PSNR, MS-SSIM) as HIP reductions.

    u8 = to_uint8(x)                      # [B,3,H,W] float in [-1,1] (or the sampler's NHWC state) -> uint8 [B,H,W,3] on the device
    write_image("out.png", u8[0])         # .png (zlib from the standard library) or .ppm (P6)
    psnr(a, b), ms_ssim(a, b)             # per-image, float32 tensors [B], computed by libcdx.so

The reference snapshot defines no image format or metric tooling (README.md: 0 bytes): PPM / PNG are the two formats every
viewer reads, the metric definitions are the standard ones (oracle/metrics_ref.py restates them with stock torch).
"""
from __future__ import annotations

import struct
import zlib

import numpy as np
import torch

from . import _abi
from .ops import _call, _ws


def to_uint8(x: torch.Tensor, *, nhwc_channels: int | None = None, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    """float32 image batch -> uint8 [B, H, W, C] (device).  x: NCHW [B,C,H,W], or -- with nhwc_channels=C -- an NHWC buffer
    [B,H,W,ld] whose first C channels are the image (the sampler's state buffer: no layout copy)."""
    assert x.is_cuda and x.dtype == torch.float32
    if nhwc_channels is None:
        x = x.permute(0, 2, 3, 1).contiguous()
        nhwc_channels = x.shape[-1]
    B, H, W, ld = x.shape
    out = torch.empty(B, H, W, nhwc_channels, dtype=torch.uint8, device=x.device)
    a = _abi.ExportU8Args(x.data_ptr(), ld, B, H * W, nhwc_channels, lo, hi, out.data_ptr())
    _call("export_u8", a, None, 0, x)
    return out


def _png_chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_png(img: np.ndarray, level: int = 6) -> bytes:
    """8-bit grey / RGB image [H,W] or [H,W,1|3] -> PNG bytes (colour type 0 / 2, filter 0 on every scanline, one IDAT)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, c = img.shape
    if c not in (1, 3):
        raise ValueError(f"PNG writer takes 1 or 3 channels, got {c}")
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, w * c)], axis=1).tobytes()      # filter byte 0 per row
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 0 if c == 1 else 2, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", ihdr) + _png_chunk(b"IDAT", zlib.compress(raw, level)) + _png_chunk(b"IEND", b"")


def encode_ppm(img: np.ndarray) -> bytes:
    """8-bit image [H,W,3] -> binary PPM (P6); [H,W] / [H,W,1] -> PGM (P5)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, c = img.shape
    if c not in (1, 3):
        raise ValueError(f"PPM writer takes 1 or 3 channels, got {c}")
    return (b"P6" if c == 3 else b"P5") + f"\n{w} {h}\n255\n".encode() + img.tobytes()


def write_image(path: str, img) -> None:
    """Write one uint8 image [H,W,C] (tensor or array) as .png or .ppm / .pgm, by extension."""
    arr = img.detach().cpu().numpy() if isinstance(img, torch.Tensor) else np.asarray(img)
    p = str(path).lower()
    if p.endswith(".png"):
        data = encode_png(arr)
    elif p.endswith((".ppm", ".pgm", ".pnm")):
        data = encode_ppm(arr)
    else:
        raise ValueError(f"write_image: unknown extension in {path!r} (use .png or .ppm)")
    with open(path, "wb") as f:
        f.write(data)


def write_images(path: str, batch_u8) -> list:
    """Write a batch [B,H,W,C]: `path` as given for one image, otherwise name_0000.ext, name_0001.ext, ...  Returns the paths."""
    B = batch_u8.shape[0]
    if B == 1:
        write_image(path, batch_u8[0])
        return [str(path)]
    stem, dot, ext = str(path).rpartition(".")
    paths = [f"{stem}_{i:04d}{dot}{ext}" for i in range(B)]
    host = batch_u8.detach().cpu() if isinstance(batch_u8, torch.Tensor) else batch_u8
    for i, pth in enumerate(paths):
        write_image(pth, host[i])
    return paths


def psnr(a: torch.Tensor, b: torch.Tensor, data_range: float = 2.0) -> torch.Tensor:
    """Per-image PSNR [B] (float32, device) of two equally shaped float32 tensors [B, ...]; data_range 2 for [-1, 1]."""
    assert a.shape == b.shape and a.is_cuda and b.is_cuda and a.dtype == b.dtype == torch.float32
    a, b = a.contiguous(), b.contiguous()
    B = a.shape[0]
    out = torch.empty(B, device=a.device)
    args = _abi.PsnrArgs(a.data_ptr(), b.data_ptr(), B, a.numel() // B, data_range, out.data_ptr())
    wp, wb, keep = _ws(_abi.workspace_bytes("psnr_f32", args), a.device)
    _call("psnr_f32", args, wp, wb, a)
    return out


def ms_ssim(x: torch.Tensor, y: torch.Tensor, data_range: float = 2.0, return_scales: bool = False):
    """Per-image MS-SSIM [B] of NCHW float32 tensors (5 scales: H, W >= 176); optionally also the five per-scale means [B,5]."""
    assert x.shape == y.shape and x.dim() == 4 and x.is_cuda and y.is_cuda and x.dtype == y.dtype == torch.float32
    x, y = x.contiguous(), y.contiguous()
    B, C, H, W = x.shape
    if min(H, W) < 176:
        raise ValueError(f"MS-SSIM with 5 scales needs images of at least 176 x 176, got {H} x {W}")
    out = torch.empty(B, device=x.device)
    scales = torch.empty(B, 5, device=x.device) if return_scales else None
    args = _abi.MsssimArgs(x.data_ptr(), y.data_ptr(), B, C, H, W, data_range, out.data_ptr(), None if scales is None else scales.data_ptr())
    wp, wb, keep = _ws(_abi.workspace_bytes("msssim_f32", args), x.device)
    _call("msssim_f32", args, wp, wb, x)
    return (out, scales) if return_scales else out
