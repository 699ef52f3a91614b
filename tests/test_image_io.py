"""(f4) Image side on the CPU: the product's PPM / PNG writers round-trip through independent readers (oracle/metrics_ref.py),
the PNG is a well-formed file (signature, chunk CRCs, IHDR fields, zlib stream), and the oracle's own metric definitions hold
their textbook identities.  No GPU needed."""
import struct
import zlib

import numpy as np
import pytest
import torch

import cdx
import oracle


@pytest.mark.parametrize("h,w,c", [(1, 1, 3), (7, 5, 3), (16, 33, 1), (64, 64, 3)])
def test_png_and_ppm_round_trip(tmp_path, h, w, c):
    rng = np.random.default_rng(h * 100 + w)
    img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    png, ppm = cdx.encode_png(img), cdx.encode_ppm(img)
    assert np.array_equal(oracle.read_png_ref(png), img) and np.array_equal(oracle.read_ppm_ref(ppm), img)
    assert png[:8] == b"\x89PNG\r\n\x1a\n" and png[12:16] == b"IHDR" and png[-8:-4] == b"IEND"
    assert struct.unpack(">IIBBBBB", png[16:29]) == (w, h, 8, 2 if c == 3 else 0, 0, 0, 0)
    assert ppm.startswith((b"P6" if c == 3 else b"P5") + f"\n{w} {h}\n255\n".encode()) and len(ppm) == len(f"P6\n{w} {h}\n255\n") + h * w * c
    for ext, rd in ((".png", oracle.read_png_ref), (".ppm", oracle.read_ppm_ref)):
        path = tmp_path / f"img{ext}"
        cdx.write_image(str(path), torch.from_numpy(img))
        assert np.array_equal(rd(path.read_bytes()), img)
    with pytest.raises(ValueError):
        cdx.write_image(str(tmp_path / "img.jpg"), img)
    with pytest.raises(ValueError):
        cdx.encode_png(np.zeros((4, 4, 2), np.uint8))


def test_write_images_names_a_batch(tmp_path):
    batch = torch.arange(3 * 4 * 5 * 3, dtype=torch.uint8).reshape(3, 4, 5, 3)
    paths = cdx.write_images(str(tmp_path / "dec.png"), batch)
    assert [p.rsplit("/", 1)[1] for p in paths] == ["dec_0000.png", "dec_0001.png", "dec_0002.png"]
    for i, p in enumerate(paths):
        assert np.array_equal(oracle.read_png_ref(open(p, "rb").read()), batch[i].numpy())
    assert cdx.write_images(str(tmp_path / "one.ppm"), batch[:1]) == [str(tmp_path / "one.ppm")]


def test_png_reader_undoes_every_filter_type():
    """The independent reader is a real decoder (all five filter types), not a mirror of the writer's filter-0-only output."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (6, 7, 3), dtype=np.uint8)
    rows, prev = [], np.zeros(21, np.int64)
    for r in range(6):
        line = img[r].reshape(-1).astype(np.int64)
        f = r % 5
        left = np.concatenate([np.zeros(3, np.int64), line[:-3]])
        ul = np.concatenate([np.zeros(3, np.int64), prev[:-3]])
        if f == 0:
            enc = line
        elif f == 1:
            enc = line - left
        elif f == 2:
            enc = line - prev
        elif f == 3:
            enc = line - (left + prev) // 2
        else:
            p = left + prev - ul
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
            enc = line - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
        rows.append(bytes([f]) + (enc & 255).astype(np.uint8).tobytes())
        prev = line
    ch = lambda t, d: struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)      # noqa: E731
    png = b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", 7, 6, 8, 2, 0, 0, 0)) + ch(b"IDAT", zlib.compress(b"".join(rows))) + ch(b"IEND", b"")
    assert np.array_equal(oracle.read_png_ref(png), img)


def test_oracle_metric_identities():
    g = torch.Generator().manual_seed(1)
    x = torch.rand(2, 3, 192, 200, generator=g) * 2 - 1
    y = (x + 0.1 * torch.randn(x.shape, generator=g)).clamp(-1, 1)
    assert torch.allclose(oracle.msssim_ref(x, x), torch.ones(2, dtype=torch.float64))
    m = oracle.msssim_ref(x, y)
    assert ((m > 0.5) & (m < 1.0)).all() and torch.allclose(m, oracle.msssim_ref(y, x))            # symmetric
    m2 = oracle.msssim_ref(x, (x + 0.3 * torch.randn(x.shape, generator=g)).clamp(-1, 1))
    assert (m2 < m).all()                                                                           # more noise, lower score
    # PSNR of a constant offset d on range 2: 10 log10(4 / d^2)
    assert torch.allclose(oracle.psnr_ref(x, x + 0.01), torch.full((2,), 10 * np.log10(4 / 1e-4), dtype=torch.float64), atol=1e-4)
    # 8-bit quantisation: endpoints, midpoint rounding, clamping
    t = torch.tensor([-1.0, 1.0, 0.0, -2.0, 3.0, -1 + 2 / 255 * 0.5, -1 + 2 / 255 * 0.49]).reshape(1, 1, 1, 7)
    assert oracle.to_uint8_ref(t).flatten().tolist() == [0, 255, 128, 0, 255, 1, 0]
