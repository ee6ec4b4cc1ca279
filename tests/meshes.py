"""Seeded synthetic meshes for builder / traversal parity tests (positions, normals|None, tex|None, triangles)."""
import numpy as np


def _soup(n, seed, extent=4.0, size=0.25):
    rng = np.random.default_rng(seed)
    c = (rng.random((n, 1, 3), dtype=np.float32) - 0.5) * extent
    pos = (c + (rng.random((n, 3, 3), dtype=np.float32) - 0.5) * size).reshape(-1, 3).astype(np.float32)
    tri = np.arange(n * 3, dtype=np.uint32).reshape(n, 3)
    return pos, None, None, tri


def _grid(n, seed):
    """Height field n x n quads -> 2 n^2 triangles with smooth normals and uv."""
    rng = np.random.default_rng(seed)
    xs, ys = np.meshgrid(np.linspace(-2, 2, n + 1, dtype=np.float32), np.linspace(-2, 2, n + 1, dtype=np.float32))
    z = (0.3 * np.sin(2.1 * xs) * np.cos(1.7 * ys) + 0.02 * rng.random(xs.shape, dtype=np.float32)).astype(np.float32)
    pos = np.stack([xs, z, ys], -1).reshape(-1, 3).astype(np.float32)
    nrm = np.stack([-0.63 * np.cos(2.1 * xs) * np.cos(1.7 * ys), np.ones_like(xs), 0.51 * np.sin(2.1 * xs) * np.sin(1.7 * ys)], -1)
    nrm = (nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)).reshape(-1, 3).astype(np.float32)
    tex = np.stack([(xs + 2) / 4, (ys + 2) / 4, np.zeros_like(xs)], -1).reshape(-1, 3).astype(np.float32)
    idx = np.arange((n + 1) * (n + 1), dtype=np.uint32).reshape(n + 1, n + 1)
    a, b, c, d = idx[:-1, :-1], idx[:-1, 1:], idx[1:, :-1], idx[1:, 1:]
    tri = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([b, d, c], -1).reshape(-1, 3)]).astype(np.uint32)
    return pos, nrm, tex, tri


def _sphere(n):
    th = np.linspace(0, np.pi, n + 1, dtype=np.float32)
    ph = np.linspace(0, 2 * np.pi, 2 * n + 1, dtype=np.float32)
    T, P = np.meshgrid(th, ph, indexing="ij")
    pos = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1).reshape(-1, 3).astype(np.float32)
    nrm = pos.copy()
    tex = np.stack([P / (2 * np.pi), T / np.pi, np.zeros_like(T)], -1).reshape(-1, 3).astype(np.float32)
    idx = np.arange(pos.shape[0], dtype=np.uint32).reshape(n + 1, 2 * n + 1)
    a, b, c, d = idx[:-1, :-1], idx[:-1, 1:], idx[1:, :-1], idx[1:, 1:]
    tri = np.concatenate([np.stack([a, c, b], -1).reshape(-1, 3), np.stack([b, c, d], -1).reshape(-1, 3)]).astype(np.uint32)
    # drop degenerate pole triangles (two identical positions) -- keeps the set non-trivial but valid
    return pos, nrm, tex, tri


def make(name):
    if name == "soup_300":
        return _soup(300, 11)
    if name == "soup_5000":
        return _soup(5000, 12, extent=10.0, size=0.2)
    if name == "grid_40":
        return _grid(40, 13)
    if name == "sphere_24":
        return _sphere(24)
    if name == "flat_plane":
        # axis-aligned plane: the root box has zero extent in y (NaN path of compress, compressed_geometry.rs:25-46)
        # 50 triangles = a single leaf (a larger planar mesh is unbuildable: see flat_plane_big)
        pos, nrm, tex, tri = _grid(5, 14)
        pos = pos.copy()
        pos[:, 1] = 0.5
        return pos, None, tex, tri
    if name == "flat_plane_big":
        # zero-volume centroid box -> bin_size 0 -> the reference's BinGrid panics (building.rs:424-429)
        pos, nrm, tex, tri = _grid(12, 14)
        pos = pos.copy()
        pos[:, 1] = 0.5
        return pos, None, tex, tri
    if name == "two_clusters":
        p1, _, _, t1 = _soup(200, 15, extent=1.0, size=0.1)
        p2, _, _, t2 = _soup(150, 16, extent=1.0, size=0.1)
        p2 = p2 + np.array([50.0, 3.0, -20.0], np.float32)
        return np.concatenate([p1, p2]).astype(np.float32), None, None, np.concatenate([t1, t2 + p1.shape[0]]).astype(np.uint32)
    if name == "sliver_fan":
        # long thin triangles sharing an apex: many coincident centroids per bin, exact-tie candidates on shared edges
        n = 400
        ang = np.linspace(0, 2 * np.pi, n + 1, dtype=np.float32)
        rim = np.stack([3 * np.cos(ang), 0.4 * np.sin(3 * ang), 3 * np.sin(ang)], -1)
        pos = np.concatenate([np.array([[0, 1.5, 0]], np.float32), rim]).astype(np.float32)
        tri = np.stack([np.zeros(n, np.uint32), np.arange(1, n + 1, dtype=np.uint32), np.arange(2, n + 2, dtype=np.uint32)], -1)
        nrm = np.tile(np.array([[0, 1, 0]], np.float32), (pos.shape[0], 1))
        return pos, nrm, None, tri.astype(np.uint32)
    raise KeyError(name)


def random_rays(n, seed, bmin, bmax):
    """Rays aimed at random points of the (slightly enlarged) box from random outside/inside origins."""
    rng = np.random.default_rng(seed)
    bmin, bmax = np.asarray(bmin, np.float32), np.asarray(bmax, np.float32)
    ext = bmax - bmin
    ctr = (bmin + bmax) / 2
    o = (ctr + (rng.random((n, 3), dtype=np.float32) - 0.5) * ext * 4.0).astype(np.float32)
    tgt = (bmin - 0.05 * ext + rng.random((n, 3), dtype=np.float32) * ext * 1.1).astype(np.float32)
    d = (tgt - o).astype(np.float32)
    # a few axis-parallel and zero-component directions (inv_direction = +inf path, geometry/mod.rs:47)
    k = max(1, n // 50)
    d[:k, 0] = 0.0
    d[k : 2 * k, 1] = -0.0
    d[2 * k : 3 * k, :2] = 0.0
    d[2 * k : 3 * k, 2] = 1.0
    return o, d
