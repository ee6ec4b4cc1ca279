"""Image output.  The reference renders into an `image::RgbaImage` and never writes it
(src/cli.rs:33-46 drops the result); this is the missing last step: RGBA8 PNG, float PFM and float OpenEXR writers, stdlib + numpy only."""
from __future__ import annotations

import struct
import zlib

import numpy as np


def save_png(path: str, rgba_u8: np.ndarray) -> None:
    """Write an [h, w, 4] uint8 image (RenderProgress.image()) as an 8-bit RGBA PNG."""
    img = np.ascontiguousarray(rgba_u8, dtype=np.uint8)
    if img.ndim != 3 or img.shape[2] != 4:
        raise ValueError("expected an [h, w, 4] uint8 image")
    h, w, _ = img.shape
    raw = np.zeros((h, w * 4 + 1), np.uint8)  # filter type 0 per scanline
    raw[:, 1:] = img.reshape(h, w * 4)

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
        f.write(chunk(b"IEND", b""))


def load_png_rgba8(path: str) -> np.ndarray:
    """Reader for the files save_png writes (filter 0, 8-bit RGBA, single IDAT stream): round-trip tests."""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        (n,) = struct.unpack(">I", data[pos:pos + 4])
        tag = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            if depth != 8 or ctype != 6:
                raise ValueError("only 8-bit RGBA")
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * 4 + 1)
    if np.any(raw[:, 0] != 0):
        raise ValueError("only filter type 0")
    return raw[:, 1:].reshape(h, w, 4).copy()


def save_pfm(path: str, rgb_f32: np.ndarray) -> None:
    """Write the pre-quantisation f32 means (RenderProgress.image_f32()[..., :3]) as a little-endian PFM."""
    img = np.ascontiguousarray(rgb_f32[..., :3], dtype="<f4")
    h, w, _ = img.shape
    with open(path, "wb") as f:
        f.write(f"PF\n{w} {h}\n-1.0\n".encode())
        f.write(img[::-1].tobytes())  # PFM stores the bottom scanline first


# ---- checkpoint / resume of a progressive render (SURVEY "aux subsystems": the reference has none; BASELINE configs[4]) ----
_CKPT_KEYS = ("tile_size", "sample_count", "width", "height", "seed", "max_depth", "chunked_sum")


def save_exr(path: str, rgba_f32: np.ndarray) -> None:
    """Write the f32 means (RenderProgress.image_f32(), [h, w, 4] or [h, w, 3]) as an uncompressed 32-bit-float scanline
    OpenEXR file (version 2, single part, channels A/B/G/R or B/G/R, INCREASING_Y): the values survive bit for bit."""
    img = np.ascontiguousarray(rgba_f32, dtype="<f4")
    if img.ndim != 3 or img.shape[2] not in (3, 4):
        raise ValueError("expected an [h, w, 3|4] float image")
    h, w, nc = img.shape
    names = ["A", "B", "G", "R"] if nc == 4 else ["B", "G", "R"]  # channels are stored in alphabetical order
    plane = {"R": 0, "G": 1, "B": 2, "A": 3}

    def attr(name: str, typ: str, value: bytes) -> bytes:
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(value)) + value

    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for n in names) + b"\0"  # 2 = FLOAT
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    header = (struct.pack("<ii", 20000630, 2)
              + attr("channels", "chlist", chlist)
              + attr("compression", "compression", b"\0")
              + attr("dataWindow", "box2i", box)
              + attr("displayWindow", "box2i", box)
              + attr("lineOrder", "lineOrder", b"\0")
              + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
              + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0))
              + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
              + b"\0")
    row_bytes = 8 + len(names) * w * 4
    first = len(header) + 8 * h
    with open(path, "wb") as f:
        f.write(header)
        f.write(np.arange(h, dtype="<u8").__mul__(row_bytes).__add__(first).tobytes())  # scanline offset table
        for y in range(h):
            f.write(struct.pack("<ii", y, len(names) * w * 4))
            for n in names:
                f.write(img[y, :, plane[n]].tobytes())


def load_exr_f32(path: str) -> np.ndarray:
    """Reader for the files save_exr writes (uncompressed FLOAT scanlines): round-trip tests.  Returns [h, w, channels] in
    R, G, B(, A) order."""
    data = open(path, "rb").read()
    if struct.unpack("<i", data[:4])[0] != 20000630:
        raise ValueError("not an OpenEXR file")
    pos, attrs = 8, {}
    while data[pos] != 0:
        e = data.index(b"\0", pos); name = data[pos:e].decode(); pos = e + 1
        e = data.index(b"\0", pos); pos = e + 1
        (n,) = struct.unpack("<i", data[pos:pos + 4]); pos += 4
        attrs[name] = data[pos:pos + n]; pos += n
    pos += 1
    if attrs["compression"] != b"\0":
        raise ValueError("only uncompressed files")
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    names, c = [], attrs["channels"]
    q = 0
    while c[q] != 0:
        e = c.index(b"\0", q); names.append(c[q:e].decode())
        if struct.unpack("<i", c[e + 1:e + 5])[0] != 2:
            raise ValueError("only FLOAT channels")
        q = e + 17
    offs = np.frombuffer(data, "<u8", h, pos)
    out = np.zeros((h, w, len(names)), np.float32)
    order = {"R": 0, "G": 1, "B": 2, "A": 3}
    for y in range(h):
        o = int(offs[y]) + 8
        for k, n in enumerate(names):
            out[y, :, order[n]] = np.frombuffer(data, "<f4", w, o + k * w * 4)
    return out


def _settings_key(settings) -> np.ndarray:
    w, h = settings.resolution
    return np.array([settings.tile_size, settings.sample_count, w, h, int(settings.seed) & 0xFFFFFFFFFFFFFFFF, settings.max_depth,
                     1 if getattr(settings, "chunked_sum", False) else 0],
                    dtype=np.uint64)


def save_checkpoint(path: str, renderer, next_sample: int) -> None:
    """Dump a FrameRenderer's running per-pixel sums (tile-major f32 RGBA: rgb = sequential sample sum, a = hit count) and the
    index of the next sample to draw, after render_pass().  Plain .npz (no pickle)."""
    tiles = np.array([[t.min_x, t.min_y, t.max_x, t.max_y] for t in renderer.tiles], dtype=np.uint32).reshape(-1, 4)
    np.savez(path, sums=renderer.tile_buf.detach().cpu().numpy(), next_sample=np.uint32(next_sample),
             settings=_settings_key(renderer.settings), tiles=tiles)


def load_checkpoint(path: str, renderer) -> int:
    """Restore the sums into `renderer.tile_buf` and return the next sample index.  The renderer must have been built for the
    same settings and tile list (the sample streams are keyed by them)."""
    import torch

    with np.load(path, allow_pickle=False) as z:
        if not np.array_equal(z["settings"], _settings_key(renderer.settings)):
            raise ValueError("checkpoint was written for other settings: " + ", ".join(_CKPT_KEYS))
        tiles = np.array([[t.min_x, t.min_y, t.max_x, t.max_y] for t in renderer.tiles], dtype=np.uint32).reshape(-1, 4)
        if not np.array_equal(z["tiles"], tiles):
            raise ValueError("checkpoint was written for another tile list")
        sums = z["sums"]
        if tuple(sums.shape) != tuple(renderer.tile_buf.shape):
            raise ValueError("checkpoint buffer shape differs")
        renderer.tile_buf.copy_(torch.from_numpy(np.ascontiguousarray(sums)))
        return int(z["next_sample"])
