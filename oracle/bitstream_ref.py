"""Oracle for the bitstream side (f4): quantisation, frequency tables, a rANS ENCODER and a decoder in plain numpy /
Python loops (small cases only).  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED by the reference (it holds no bitstream
format: README.md is 0 bytes); the algorithm is static-model rANS as published (J. Duda, "Asymmetric numeral systems",
2013; word-renormalising 32-bit form), the container is the build-defined "CDXL" v1 documented in
conditional-diffusion-model-for-compression_amd/bitstream.py -- restated here independently of it.
"""
from __future__ import annotations

import struct

import numpy as np

L = 1 << 16


def quantise_ref(z: np.ndarray, step: float, qmax: int) -> np.ndarray:
    return np.clip(np.rint(np.asarray(z, np.float64) / step), -qmax, qmax).astype(np.int64)


def build_freq_ref(symbols: np.ndarray, qmax: int, prob_bits: int) -> np.ndarray:
    """Frequencies proportional to the histogram, every symbol >= 1, sum exactly 2^prob_bits."""
    A, M = 2 * qmax + 1, 1 << prob_bits
    hist = np.bincount((np.asarray(symbols).ravel() + qmax).astype(np.int64), minlength=A).astype(np.float64) + 1e-9
    f = np.maximum(1, np.floor(hist / hist.sum() * (M - A)).astype(np.int64) + 1)
    while f.sum() > M:
        f[np.argmax(f)] -= 1
    while f.sum() < M:
        f[np.argmax(hist - f / M * hist.sum())] += 1
    assert f.sum() == M and f.min() >= 1
    return f.astype(np.uint16)


def rans_encode_ref(sym: np.ndarray, freq: np.ndarray, prob_bits: int) -> list[int]:
    """symbols in [0, alphabet) -> stream words: [state >> 16, state & 0xffff, renormalisation words in decode order]."""
    cum = np.concatenate([[0], np.cumsum(freq.astype(np.int64))])
    x, out = L, []
    for s in reversed([int(v) for v in sym]):
        f, c = int(freq[s]), int(cum[s])
        x_max = ((L >> prob_bits) << 16) * f
        while x >= x_max:
            out.append(x & 0xFFFF)
            x >>= 16
        x = ((x // f) << prob_bits) + (x % f) + c
    return [x >> 16, x & 0xFFFF] + out[::-1]


def rans_decode_ref(words, nsym: int, freq: np.ndarray, prob_bits: int) -> np.ndarray:
    cum = np.concatenate([[0], np.cumsum(freq.astype(np.int64))])
    slot2sym = np.repeat(np.arange(freq.size), freq.astype(np.int64))
    x, pos, out = (int(words[0]) << 16) | int(words[1]), 2, []
    for _ in range(nsym):
        slot = x & ((1 << prob_bits) - 1)
        s = int(slot2sym[slot])
        x = int(freq[s]) * (x >> prob_bits) + slot - int(cum[s])
        if x < L:
            x = (x << 16) | int(words[pos])
            pos += 1
        out.append(s)
    assert x == L and pos == len(words), "stream does not end in the initial state"
    return np.asarray(out, np.int64)


def encode_latent_ref(q: np.ndarray, step: float, qmax: int, prob_bits: int = 12, freq: np.ndarray | None = None) -> bytes:
    """q [Cz, h, w] integers in [-qmax, qmax] -> one CDXL v1 container."""
    cz, h, w = q.shape
    freq = build_freq_ref(q, qmax, prob_bits) if freq is None else freq
    streams = [rans_encode_ref(q[c].ravel() + qmax, freq, prob_bits) for c in range(cz)]
    off, ln, payload = [], [], []
    for st in streams:
        off.append(len(payload))
        ln.append(len(st))
        payload += st
    A = 2 * qmax + 1
    buf = b"CDXL" + struct.pack("<6H", 1, prob_bits, cz, h, w, qmax) + struct.pack("<fI", step, len(payload))
    buf += np.asarray(list(freq) + [0] * (A & 1), "<u2").tobytes()
    buf += np.asarray(off, "<u4").tobytes() + np.asarray(ln, "<u4").tobytes() + np.asarray(payload, "<u2").tobytes()
    return buf


def decode_latent_ref(buf: bytes):
    """One CDXL v1 container -> (symbols [Cz, h, w] int64 in [-qmax, qmax], z float32 = symbols * step)."""
    assert buf[:4] == b"CDXL"
    version, pb, cz, h, w, qmax = struct.unpack_from("<6H", buf, 4)
    step, nwords = struct.unpack_from("<fI", buf, 16)
    A = 2 * qmax + 1
    pos = 24
    freq = np.frombuffer(buf, "<u2", A, pos)
    pos += 2 * (A + (A & 1))
    off = np.frombuffer(buf, "<u4", cz, pos)
    ln = np.frombuffer(buf, "<u4", cz, pos + 4 * cz)
    words = np.frombuffer(buf, "<u2", nwords, pos + 8 * cz)
    q = np.stack([rans_decode_ref(words[off[c]:off[c] + ln[c]], h * w, freq, pb).reshape(h, w) - qmax for c in range(cz)])
    return q, (q.astype(np.float32) * np.float32(step))
