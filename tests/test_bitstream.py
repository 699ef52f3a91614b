"""(f4) Bitstream side on the CPU: the oracle's rANS encoder / decoder round-trip and agree with the committed golden
container, and the product's container parser accepts exactly what the format allows.  No GPU needed."""
import os
import struct

import numpy as np
import pytest

import cdx
import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "latent_stream.npz")


@pytest.mark.parametrize("cz,h,w,qmax,pb,scale", [(4, 6, 5, 15, 12, 1.0), (1, 1, 1, 3, 8, 1.0), (3, 16, 16, 127, 12, 20.0), (2, 7, 3, 0, 4, 1.0),
                                                  (5, 8, 8, 31, 10, 0.05), (2, 32, 32, 15, 12, 3.0)])
def test_oracle_rans_round_trip(cz, h, w, qmax, pb, scale):
    rng = np.random.default_rng(cz * 100 + h)
    q = oracle.quantise_ref(rng.standard_normal((cz, h, w)) * scale, 0.5, qmax)
    buf = oracle.encode_latent_ref(q, 0.5, qmax, pb)
    q2, z2 = oracle.decode_latent_ref(buf)
    assert np.array_equal(q, q2) and np.array_equal(z2, q.astype(np.float32) * np.float32(0.5))
    p = cdx.parse_latent_stream(buf)
    assert (p["channels"], p["height"], p["width"], p["qmax"], p["prob_bits"]) == (cz, h, w, qmax, pb)
    assert int(p["freq"].astype(np.int64).sum()) == 1 << pb and p["freq"].min() >= 1
    # coding efficiency: payload within the ideal code length of the table + 4 bytes of state and 2 of slack per stream
    f = p["freq"].astype(np.float64) / (1 << pb)
    ideal_bits = -np.log2(f[(q + qmax).ravel()]).sum()
    assert p["words"].size * 16 <= ideal_bits + cz * 48


def test_golden_container_decodes_to_golden_symbols():
    g = np.load(GOLD)
    q, z = oracle.decode_latent_ref(g["container"].tobytes())
    assert np.array_equal(q, g["symbols"]) and np.array_equal(z, g["symbols"].astype(np.float32) * g["step"])
    # the encoder is deterministic: re-encoding the golden symbols gives the golden bytes
    assert oracle.encode_latent_ref(g["symbols"].astype(np.int64), float(g["step"]), 7) == g["container"].tobytes()


def test_parser_rejects_malformed_containers():
    good = np.load(GOLD)["container"].tobytes()
    cdx.parse_latent_stream(good)
    for bad in (b"XXXX" + good[4:], good[:-2], good + b"\0\0", good[:4] + struct.pack("<H", 2) + good[6:],
                good[:6] + struct.pack("<H", 13) + good[8:]):
        with pytest.raises(ValueError):
            cdx.parse_latent_stream(bad)
    broken = bytearray(good)
    broken[24] ^= 1                                         # frequency table no longer sums to 2^prob_bits
    with pytest.raises(ValueError, match="frequency"):
        cdx.parse_latent_stream(bytes(broken))
    p = cdx.parse_latent_stream(good)
    pos = 24 + 2 * (15 + 1)
    broken = bytearray(good)
    broken[pos:pos + 4] = struct.pack("<I", 10 ** 6)        # stream offset outside the payload
    with pytest.raises(ValueError, match="outside"):
        cdx.parse_latent_stream(bytes(broken))
    assert p["words"].size == struct.unpack_from("<I", good, 20)[0]


def test_parser_bounds_what_a_tiny_container_can_request():
    """a ~40-byte hostile container must not be able to ask for 2^31 symbols per stream."""
    good = np.load(GOLD)["container"].tobytes()
    huge = bytearray(good)
    struct.pack_into("<2H", huge, 10, 65535, 65535)          # h, w (header: magic, version, pb, cz, h, w, qmax)
    with pytest.raises(ValueError, match="limits"):
        cdx.parse_latent_stream(bytes(huge))
    # within the side limits, but far more symbols than the payload's bits can encode
    p = cdx.parse_latent_stream(good)
    many = bytearray(good)
    struct.pack_into("<2H", many, 10, 1024, 1024)
    with pytest.raises(ValueError, match="too short"):
        cdx.parse_latent_stream(bytes(many))
    assert p["height"] * p["width"] < 1024 * 1024
