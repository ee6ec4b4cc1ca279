import numpy as np
import pytest

from minipath_amd import io


def test_png_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    p = str(tmp_path / "a.png")
    io.save_png(p, img)
    assert np.array_equal(io.load_png_rgba8(p), img)
    assert open(p, "rb").read(8) == b"\x89PNG\r\n\x1a\n"


def test_pfm_header(tmp_path):
    img = np.linspace(0, 1, 3 * 4 * 4, dtype=np.float32).reshape(3, 4, 4)
    p = str(tmp_path / "a.pfm")
    io.save_pfm(p, img)
    data = open(p, "rb").read()
    assert data.startswith(b"PF\n4 3\n-1.0\n")
    body = np.frombuffer(data[len(b"PF\n4 3\n-1.0\n"):], "<f4").reshape(3, 4, 3)
    assert np.array_equal(body[::-1], img[..., :3])


def test_exr_round_trip_keeps_every_bit(tmp_path):
    """save_exr: uncompressed FLOAT scanlines; the f32 means (incl. inf, NaN payloads, -0, subnormals) come back bit for bit, and
    the file carries the attributes every OpenEXR reader requires."""
    rng = np.random.default_rng(5)
    for shape in ((7, 13, 4), (5, 3, 3), (1, 1, 4)):
        img = rng.standard_normal(shape).astype(np.float32)
        flat = img.reshape(-1)
        flat[:5] = np.array([np.inf, -0.0, 1e-42, -np.inf, 0.0], np.float32)[: min(5, flat.size)]
        if flat.size > 6:
            flat[6:7] = np.array([0x7FC12345], np.uint32).view(np.float32)
        p = str(tmp_path / "a.exr")
        io.save_exr(p, img)
        back = io.load_exr_f32(p)
        assert back.shape == shape and np.array_equal(back.view(np.uint32), img.view(np.uint32))
        raw = open(p, "rb").read()
        assert raw[:8] == bytes([0x76, 0x2F, 0x31, 0x01, 2, 0, 0, 0])
        for name in (b"channels\0chlist\0", b"compression\0compression\0", b"dataWindow\0box2i\0", b"displayWindow\0box2i\0",
                     b"lineOrder\0lineOrder\0", b"pixelAspectRatio\0float\0", b"screenWindowCenter\0v2f\0", b"screenWindowWidth\0float\0"):
            assert name in raw
        h, w, nc = shape
        hdr_end = raw.index(b"screenWindowWidth\0float\0") + len(b"screenWindowWidth\0float\0") + 4 + 4 + 1
        assert len(raw) == hdr_end + 8 * h + h * (8 + nc * w * 4)
    with pytest.raises(ValueError):
        io.save_exr(str(tmp_path / "b.exr"), np.zeros((4, 4), np.float32))
