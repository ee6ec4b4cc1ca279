import numpy as np

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
