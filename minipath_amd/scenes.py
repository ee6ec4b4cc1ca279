"""Procedural scenes for benchmarks (numpy only).

`atrium()` is the seeded stand-in for the Sponza configs of BASELINE.json: the reference's `data/Sponza` is an empty
git submodule, so no Sponza mesh exists in this pipeline.  The generator emits TRIANGLES ONLY (the
reference's loader drops every other polygon, building.rs:43-46): a two-storey colonnaded hall with arches, a
vaulted roof, walls, a floor and draped cloth (fine grids, small triangles), extent about 37 x 15 x 23 units.
"""
from __future__ import annotations

import numpy as np


class _Mesh:
    def __init__(self, rng=None):
        self.pos, self.nrm, self.tex, self.tri, self.n = [], [], [], [], 0
        self.rng = rng

    def add(self, pos, nrm, tex, tri):
        pos = np.asarray(pos, np.float32).reshape(-1, 3)
        self.pos.append(pos)
        self.nrm.append(np.asarray(nrm, np.float32).reshape(-1, 3))
        self.tex.append(np.asarray(tex, np.float32).reshape(-1, 3))
        self.tri.append(np.asarray(tri, np.int64).reshape(-1, 3) + self.n)
        self.n += pos.shape[0]

    def grid(self, f, nu, nv, flip=False):
        """Parametric surface f(u,v) -> (pos, normal) on a (nu x nv)-cell grid: 2*nu*nv triangles."""
        u, v = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
        if self.rng is not None:
            # irregular tessellation: a regular grid puts whole rows of triangle centroids on one exact coordinate, and a
            # BVH node holding such a row has a zero-extent centroid box, which the reference's BinGrid cannot bin
            ju = (self.rng.random(u.shape) - 0.5) * (0.3 / nu)
            jv = (self.rng.random(v.shape) - 0.5) * (0.3 / nv)
            ju[0, :] = ju[-1, :] = 0.0
            jv[:, 0] = jv[:, -1] = 0.0
            u, v = u + ju, v + jv
        p, n = f(u, v)
        idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
        a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[:-1, 1:], idx[1:, 1:]
        t = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([b, d, c], -1).reshape(-1, 3)])
        if flip:
            t = t[:, ::-1]
        self.add(p.reshape(-1, 3), n.reshape(-1, 3), np.stack([u, v, np.zeros_like(u)], -1).reshape(-1, 3), t)

    def arrays(self):
        return (np.concatenate(self.pos).astype(np.float32), np.concatenate(self.nrm).astype(np.float32),
                np.concatenate(self.tex).astype(np.float32), np.concatenate(self.tri).astype(np.uint32))


def _norm(v):
    return v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-20)


def atrium(seed: int = 1, detail: float = 1.0):
    """Returns (positions[nv,3] f32, normals[nv,3] f32, tex[nv,3] f32, triangles[nt,3] u32).
    detail = 1.0 gives 262 k triangles (+-1 %); smaller values give proportionally lighter scenes for tests."""
    rng = np.random.default_rng(seed)
    m = _Mesh(rng)
    L, H, W = 37.0, 15.0, 23.0  # x in [-L/2, L/2], y in [0, H], z in [-W/2, W/2]
    k = float(np.sqrt(max(detail, 1e-3)))

    def n_(x):
        return max(2, int(round(x * k)))

    def plane(o, du, dv, nrm):
        """Masonry-like relief along the normal: an exactly planar axis-aligned patch of more than 56 triangles has a
        zero-volume centroid box, on which the reference's BinGrid panics (building.rs:424-429)."""
        o, du, dv, nrm = (np.asarray(a, np.float64) for a in (o, du, dv, nrm))
        ph = rng.random(2) * 6.28

        def f(u, v):
            bump = 0.03 * np.sin(u * 37.0 + ph[0]) * np.sin(v * 29.0 + ph[1]) + 0.01 * np.sin(u * 113.0) * np.sin(v * 97.0)
            p = o + u[..., None] * du + v[..., None] * dv + bump[..., None] * nrm
            return p, np.broadcast_to(nrm, p.shape)
        return f

    # floor with a gentle tiling relief, walls, gallery floors
    def floor(u, v):
        x, z = (u - 0.5) * L, (v - 0.5) * W
        y = 0.02 * np.sin(x * 6.0) * np.sin(z * 6.0)
        p = np.stack([x, y, z], -1)
        n = _norm(np.stack([-0.12 * np.cos(x * 6.0) * np.sin(z * 6.0), np.ones_like(x), -0.12 * np.sin(x * 6.0) * np.cos(z * 6.0)], -1))
        return p, n
    m.grid(floor, n_(150), n_(96))
    m.grid(plane([-L / 2, 0, -W / 2], [L, 0, 0], [0, H, 0], [0, 0, 1]), n_(96), n_(40))
    m.grid(plane([-L / 2, 0, W / 2], [L, 0, 0], [0, H, 0], [0, 0, -1]), n_(96), n_(40), flip=True)
    m.grid(plane([-L / 2, 0, -W / 2], [0, 0, W], [0, H, 0], [1, 0, 0]), n_(60), n_(40), flip=True)
    m.grid(plane([L / 2, 0, -W / 2], [0, 0, W], [0, H, 0], [-1, 0, 0]), n_(60), n_(40))
    for zs in (-1.0, 1.0):  # first-floor galleries
        z0 = zs * W / 2
        m.grid(plane([-L / 2, 6.0, z0], [L, 0, 0], [0, 0, -zs * 4.5], [0, 1, 0]), n_(96), n_(12), flip=zs > 0)
        m.grid(plane([-L / 2, 5.7, z0], [L, 0, 0], [0, 0, -zs * 4.5], [0, -1, 0]), n_(96), n_(12), flip=zs < 0)

    # barrel-vaulted roof
    def roof(u, v):
        x = (u - 0.5) * L
        a = v * np.pi
        z, y = -np.cos(a) * W / 2, 11.0 + np.sin(a) * 4.0
        n = _norm(np.stack([np.zeros_like(x), -np.sin(a) * W / 2, np.cos(a) * 4.0], -1))
        return np.stack([x, y, z], -1), n
    m.grid(roof, n_(128), n_(64))

    # columns (two storeys, both sides) and arches between them
    def column(cx, cz, y0, y1, r):
        def f(u, v):
            a = u * 2 * np.pi
            y = y0 + v * (y1 - y0)
            rr = r * (1.0 + 0.08 * np.cos(a * 12.0)) * (1.0 - 0.15 * v)  # fluting + taper
            p = np.stack([cx + rr * np.cos(a), y, cz + rr * np.sin(a)], -1)
            return p, _norm(np.stack([np.cos(a), 0.05 * np.ones_like(a), np.sin(a)], -1))
        return f

    def arch(x0, x1, cz, y0, r):
        cx, R = (x0 + x1) / 2, (x1 - x0) / 2

        def f(u, v):
            a, b = u * np.pi, v * 2 * np.pi
            ring = np.stack([cx - np.cos(a) * (R + r * np.cos(b)), y0 + np.sin(a) * (R + r * np.cos(b)), cz + r * np.sin(b)], -1)
            n = _norm(np.stack([-np.cos(a) * np.cos(b), np.sin(a) * np.cos(b), np.sin(b)], -1))
            return ring, n
        return f
    xs = np.linspace(-L / 2 + 2.0, L / 2 - 2.0, 11)
    for zs in (-1.0, 1.0):
        cz = zs * (W / 2 - 4.5)
        for storey, (y0, y1, r) in enumerate(((0.0, 5.7, 0.45), (6.0, 10.2, 0.32))):
            for x in xs:
                m.grid(column(x, cz, y0, y1, r), n_(24), n_(14))
            for xa, xb in zip(xs[:-1], xs[1:]):
                m.grid(arch(xa + r, xb - r, cz, y1 - 1.4, 0.22), n_(20), n_(8))

    # draped cloth: fine grids with folds (small triangles, like Sponza's curtains)
    for i, (x0, z0, wx, hy) in enumerate(((-12.0, -5.5, 7.0, 5.0), (-2.0, 5.5, 8.0, 5.5), (8.0, -5.5, 7.0, 5.0), (3.0, 0.0, 6.0, 4.0))):
        ph = rng.random(3) * 6.28

        def cloth(u, v, x0=x0, z0=z0, wx=wx, hy=hy, ph=ph):
            x = x0 + u * wx
            y = 10.5 - v * hy
            z = z0 + 0.35 * np.sin(u * 19.0 + ph[0]) * (0.3 + v) + 0.12 * np.sin(v * 23.0 + ph[1]) + 0.05 * np.sin((u + v) * 41.0 + ph[2])
            dzdu = 0.35 * 19.0 * np.cos(u * 19.0 + ph[0]) * (0.3 + v) + 0.05 * 41.0 * np.cos((u + v) * 41.0 + ph[2])
            dzdv = 0.35 * np.sin(u * 19.0 + ph[0]) + 0.12 * 23.0 * np.cos(v * 23.0 + ph[1]) + 0.05 * 41.0 * np.cos((u + v) * 41.0 + ph[2])
            n = _norm(np.stack([-dzdu / wx, dzdv / hy, np.ones_like(u)], -1))
            return np.stack([x, y, z], -1), n
        m.grid(cloth, n_(150), n_(110))

    # a few vases on the floor (spheres), flat-shaded (zero normals => flat, building.rs:200)
    for _ in range(6):
        c = np.array([rng.uniform(-L / 2 + 3, L / 2 - 3), 0.6, rng.uniform(-4.0, 4.0)])

        def vase(u, v, c=c):
            a, b = u * 2 * np.pi, (v - 0.5) * np.pi
            p = c + 0.6 * np.stack([np.cos(b) * np.cos(a), np.sin(b), np.cos(b) * np.sin(a)], -1)
            return p, np.zeros_like(p)
        m.grid(vase, n_(28), n_(14))
    return m.arrays()


ATRIUM_VIEW = ((-16.0, 4.2, 0.8), (12.0, 5.5, -0.5), 4.0)  # eye, target, f-number


def atrium_camera():
    """Inside the hall, looking down the nave."""
    from .camera import Camera

    eye, at, fnum = ATRIUM_VIEW
    return Camera.default().look_at(eye, at, (0.0, 1.0, 0.0)).f_number(fnum)


def write_obj(path, pos, nrm, tex, tri):
    """Triangles-only OBJ with per-vertex vn/vt (f a/a/a b/b/b c/c/c)."""
    with open(path, "w") as f:
        for p in pos:
            f.write(f"v {p[0]:.7g} {p[1]:.7g} {p[2]:.7g}\n")
        for t in tex:
            f.write(f"vt {t[0]:.7g} {t[1]:.7g}\n")
        for n in nrm:
            f.write(f"vn {n[0]:.7g} {n[1]:.7g} {n[2]:.7g}\n")
        for a, b, c in tri + 1:
            f.write(f"f {a}/{a}/{a} {b}/{b}/{b} {c}/{c}/{c}\n")
