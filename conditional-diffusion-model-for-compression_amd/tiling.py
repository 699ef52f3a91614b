"""S5: tiled decode -- images larger than the UNet's native size are cut into overlapping
tiles, the tiles are decoded as ordinary batch entries, and the results are blended with linear ramps.

Build-defined (the reference snapshot has no tiling code: README.md is 0 bytes).  Definition:
  * tile origins along an axis: 0, s, 2s, ... with s = tile - overlap, last origin clamped to size - tile;
    origins must be multiples of the conditioning stride (16) so every tile owns whole conditioning cells;
  * a tile's conditioning is the matching crop of `cond`; its x_T is the matching crop of ONE full-image noise
    field (generator stream keyed by the global image index, element index = NCHW index in the FULL image), so
    overlapping tiles start from identical noise where they overlap;
  * tiles are decoded with the deterministic DDIM sampler; the blend is cdx_tile_blend_f32.
"""
from __future__ import annotations

import torch

from . import _abi, ops
from .rng import STREAM_XT

COND_STRIDE = 16


def tile_origins(size: int, tile: int, overlap: int) -> list[int]:
    if size < tile:
        raise ValueError(f"image side {size} smaller than the tile {tile}")
    if not 0 <= overlap < tile:
        raise ValueError("overlap must be in [0, tile)")
    step = tile - overlap
    o = list(range(0, size - tile, step)) + [size - tile]
    if any(v % COND_STRIDE for v in o):
        raise ValueError(f"tile origins {o} are not multiples of the conditioning stride {COND_STRIDE}")
    return o


def tile_plan(h: int, w: int, tile: int, overlap: int) -> tuple[list[int], list[int]]:
    return tile_origins(h, tile, overlap), tile_origins(w, tile, overlap)


@torch.no_grad()
def tile_batch(net, cond: torch.Tensor, overlap: int, seed: int, first_image: int):
    """The tiles of `cond`'s images as ordinary batch entries: (conds [B*ny*nx, Cc, T/16, T/16], x_T [B*ny*nx, C, T, T],
    ys, xs).  x_T tiles are crops of ONE noise field per image (generator stream keyed by the global image index)."""
    cfg, dev = net.cfg, net.device
    B, Cc, hc, wc = cond.shape
    T, C = cfg["image_size"], cfg["in_channels"]
    H, W = hc * COND_STRIDE, wc * COND_STRIDE
    ys, xs = tile_plan(H, W, T, overlap)
    cond = cond.to(dev, torch.float32)
    full = torch.zeros(B, H, W, 4, device=dev)
    ops.gauss_fill(full, C, seed, first_image, STREAM_XT)
    full = full[..., :C].permute(0, 3, 1, 2)
    ct = T // COND_STRIDE
    conds, xts = [], []
    for b in range(B):
        for y in ys:
            for x in xs:
                conds.append(cond[b, :, y // COND_STRIDE:y // COND_STRIDE + ct, x // COND_STRIDE:x // COND_STRIDE + ct])
                xts.append(full[b, :, y:y + T, x:x + T])
    return torch.stack(conds).contiguous(), torch.stack(xts).contiguous(), ys, xs


@torch.no_grad()
def blend_tiles(net, tiles: torch.Tensor, B: int, ys: list[int], xs: list[int], H: int, W: int) -> torch.Tensor:
    """tiles [B*ny*nx, C, T, T] (decoded, device) -> [B, C, H, W] with separable linear ramps (cdx_tile_blend_f32)."""
    dev, T, C = net.device, net.cfg["image_size"], net.cfg["in_channels"]
    tiles = tiles.contiguous()
    out = torch.empty(B, C, H, W, device=dev)
    y0 = torch.tensor(ys, dtype=torch.int32, device=dev)
    x0 = torch.tensor(xs, dtype=torch.int32, device=dev)
    a = _abi.TileBlendArgs(tiles.data_ptr(), B, C, T, len(ys), len(xs), y0.data_ptr(), x0.data_ptr(), H, W, out.data_ptr())
    _abi.call("tile_blend_f32", a, None, 0, torch.cuda.current_stream(dev).cuda_stream)
    return out


@torch.no_grad()
def sample_tiled(sampler, cond: torch.Tensor, steps: int, *, overlap: int = 64, seed: int = 0, first_image: int = 0,
                 tiles_per_call: int = 16) -> torch.Tensor:
    """cond [B, Cc, hc, wc] -> x_0 [B, C, 16*hc, 16*wc] decoded through the sampler's (tile-sized) UNet."""
    net, cfg = sampler.unet, sampler.unet.cfg
    if sampler.method != "ddim" or cfg["cond_mode"] != "concat":
        raise ValueError("tiled decode is defined for the deterministic DDIM sampler with concat conditioning")
    with torch.cuda.device(net.device):
        B, _, hc, wc = cond.shape
        conds, xts, ys, xs = tile_batch(net, cond, overlap, seed, first_image)
        outs = []
        for i in range(0, conds.shape[0], tiles_per_call):
            outs.append(sampler.sample(conds[i:i + tiles_per_call], steps, seed=seed, x_T=xts[i:i + tiles_per_call]))
        return blend_tiles(net, torch.cat(outs), B, ys, xs, hc * COND_STRIDE, wc * COND_STRIDE)
