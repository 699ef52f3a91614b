"""Oracle restatement of the counter-based generator (S4).  TEST INFRASTRUCTURE ONLY.

Independent of the product's rng.py: plain Python integers (arbitrary precision masked to
64 bits) and the math module, one element at a time for the scalar form, plus a numpy
form used for whole tensors.  PARITY UNPINNED by the reference (it defines no generator).
"""
from __future__ import annotations

import math

import numpy as np

_MASK = (1 << 64) - 1
_G = 0x9E3779B97F4A7C15


def _mix(z: int) -> int:
    z &= _MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


def stream_key_ref(seed: int, a: int, b: int) -> int:
    k = _mix((seed & _MASK) + _G)
    k = _mix(k ^ ((a + _G) & _MASK))
    return _mix(k ^ ((b + _G) & _MASK))


def hash_scalar(key: int, idx: int) -> int:
    return _mix(key + (idx + 1) * _G)


def normal_scalar(key: int, idx: int) -> float:
    h = hash_scalar(key, idx)
    u1 = ((h >> 32) + 0.5) / 4294967296.0
    u2 = ((h & 0xFFFFFFFF) + 0.5) / 4294967296.0
    return float(np.float32(math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * math.pi * u2)))


def _hash_vec(key: int, n: int, offset: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint64)
    g = np.uint64(_G)
    idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(key) + idx * g
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        out[:] = z ^ (z >> np.uint64(31))
    return out


def uniform_ref(key: int, n: int, offset: int = 0) -> np.ndarray:
    h = _hash_vec(key, n, offset)
    return ((h >> np.uint64(40)).astype(np.float64) + 0.5) / float(1 << 24)


def normal_ref(key: int, n: int, offset: int = 0) -> np.ndarray:
    h = _hash_vec(key, n, offset)
    u1 = ((h >> np.uint64(32)).astype(np.float64) + 0.5) / 4294967296.0
    u2 = ((h & np.uint64(0xFFFFFFFF)).astype(np.float64) + 0.5) / 4294967296.0
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)
