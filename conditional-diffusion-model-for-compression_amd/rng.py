"""Counter-based generator (S4 of SURVEY.md section 8a), host side (numpy).

Used on the host for synthetic weights and inputs; the device restatement is
`cdx_gauss_fill_f32` / the noise term of `cdx_diffusion_update_f32` in csrc/pointwise.hip.
Definition (build-owned; the reference snapshot defines none):

    mix64(z): z ^= z >> 30; z *= 0xBF58476D1CE4E5B9; z ^= z >> 27;
              z *= 0x94D049BB133111EB; z ^= z >> 31                      (splitmix64 finaliser)
    key(seed, a, b) = mix64(mix64(mix64(seed + G) ^ (a + G)) ^ (b + G)),  G = 0x9E3779B97F4A7C15
    h(idx)          = mix64(key + (idx + 1) * G)
    uniform24(idx)  = ((h >> 40) + 0.5) * 2^-24                           in (0, 1)
    normal(idx)     = float32( sqrt(-2 ln u1) * cos(2 pi u2) ),  u1 = ((h >> 32) + 0.5) 2^-32,
                      u2 = ((h & 0xffffffff) + 0.5) 2^-32, evaluated in float64.
"""
from __future__ import annotations

import numpy as np

GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

# Stream ids (the `b` argument of key()); image streams use a = global image index.
STREAM_TARGET = 0      # synthetic ground-truth image
STREAM_XT = 1          # x_T
STREAM_COND = 2        # conditioning noise / tokens
STREAM_STEP0 = 16      # DDPM noise of reverse step k uses stream STREAM_STEP0 + k
PARAM_A = (1 << 40)    # a = PARAM_A + parameter index for weight streams


def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def stream_key(seed: int, a: int, b: int) -> int:
    with np.errstate(over="ignore"):
        k = mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + GOLD)
        k = mix64(k ^ (np.uint64(a) + GOLD))
        k = mix64(k ^ (np.uint64(b) + GOLD))
    return int(k)


def _hash(key: int, n: int, offset: int = 0) -> np.ndarray:
    idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return mix64(np.uint64(key) + idx * GOLD)


def uniform(key: int, n: int, offset: int = 0) -> np.ndarray:
    """float64 uniforms in (0,1), 24 bits."""
    h = _hash(key, n, offset)
    return ((h >> np.uint64(40)).astype(np.float64) + 0.5) * (1.0 / (1 << 24))


def normal(key: int, n: int, offset: int = 0) -> np.ndarray:
    """float32 standard normals (Box-Muller, cosine branch, float64 arithmetic)."""
    h = _hash(key, n, offset)
    u1 = ((h >> np.uint64(32)).astype(np.float64) + 0.5) * (1.0 / 4294967296.0)
    u2 = ((h & np.uint64(0xFFFFFFFF)).astype(np.float64) + 0.5) * (1.0 / 4294967296.0)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)
