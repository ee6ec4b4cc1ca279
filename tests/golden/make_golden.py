"""Generates tests/golden/teapot_golden.npz from the CPU oracle (SURVEY 8c "golden fixtures the build commits").

The Rust reference cannot run in this pipeline (no cargo/rustc), so these vectors are NOT reference outputs: they
freeze the oracle's restatement so that any later drift of the oracle or of the HIP path is caught.  The only
reference-held data file on this path, data/teapot.obj (a mesh, not source), is committed beside this script.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

SEED = 0x5EED


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    b = po.Bvh.from_obj(os.path.join(here, "teapot.obj"))
    s = po.build_sampler(po.teapot_camera(), 256, 256)
    rng = np.random.default_rng(2024)
    n = 4096
    o = np.zeros((n, 3), np.float32)
    d = np.zeros((n, 3), np.float32)
    xs, ys = rng.integers(0, 256, n), rng.integers(0, 256, n)
    for i in range(n):
        r = po.sample_ray(s, int(xs[i]), int(ys[i]), 77000 + i)
        o[i], d[i] = list(r.o), list(r.d)
    t, prim, u, v = b.trace(o, d)
    tile = (64, 96, 128, 160)
    f, u8 = b.render_tile(s, 256, 256, 16, SEED, *tile)
    bmin, bmax = b.bbox()
    digest = {
        "inner": hashlib.sha256(b.inner_nodes_bytes().tobytes()).hexdigest(),
        "packets": hashlib.sha256(b.packets_bytes().tobytes()).hexdigest(),
        "shading": hashlib.sha256(b.tri_shading().tobytes()).hexdigest(),
    }
    np.savez_compressed(
        os.path.join(here, "teapot_golden.npz"),
        seed=np.uint64(SEED), sampler=s.as_array(), ray_o=o, ray_d=d, hit_t_bits=t.view(np.uint32), hit_prim=prim,
        hit_u_bits=u.view(np.uint32), hit_v_bits=v.view(np.uint32), tile=np.array(tile, np.uint32),
        tile_f32_bits=f.view(np.uint32), tile_u8=u8,
        bvh_counts=np.array([b.n_inner, b.n_packets, b.n_vertices, b.depth, b.root], np.uint32),
        bvh_bbox=np.concatenate([bmin, bmax]),
        bvh_sha256=np.array([digest["inner"], digest["packets"], digest["shading"]]),
    )
    print("wrote teapot_golden.npz:", b.n_inner, "inner,", b.n_packets, "packets;", int((prim != po.NO_PRIM).sum()), "of", n, "rays hit")


if __name__ == "__main__":
    main()
