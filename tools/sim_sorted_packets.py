#!/usr/bin/env python3
"""Would sorting incoherent (bounce) rays into 64-ray packets pay?  Bounce-like rays leaving the atrium's surfaces are keyed by
(origin grid cell, direction bin), sorted, cut into packets of 64; for each packet the union of inner nodes / leaf packets its rays
visit (what the packet walk would pay) is compared with the per-ray average (what a single ray needs).  Diagnostics only."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po
from minipath_amd import scenes

def visits(b, o, d):
    L = po.lib(); buf = (C.c_uint8 * 16384)(); lbuf = (C.c_uint32 * 16384)()
    out = []
    for i in range(o.shape[0]):
        r = po.ray_new(o[i], d[i])
        n = L.mpo_bvh_intersect_ops(b.h, C.byref(r), buf, lbuf, 16384)
        nodes = [lbuf[k] for k in range(n) if buf[k] == 1]
        leaves = [(lbuf[k], buf[k] - 8) for k in range(n) if buf[k] >= 8]
        out.append((nodes, leaves))
    return out

def main():
    detail = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
    b = po.Bvh.build(*scenes.atrium(1, detail))
    rng = np.random.default_rng(3)
    lo = np.array([-17.0, 0.5, -10.0], np.float32); hi = np.array([17.0, 13.0, 10.0], np.float32)
    o = (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    t, prim, u, v = b.trace(o, d)
    hit = prim != 0xFFFFFFFF
    o2 = (o + d * t[:, None])[hit]
    d2 = rng.normal(size=o2.shape).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o2 = (o2 + 1e-3 * d2).astype(np.float32)
    vis = visits(b, o2, d2)
    per_nodes = np.mean([len(a) for a, _ in vis]); per_pk = np.mean([sum(k for _, k in l) for _, l in vis])
    print(f"{o2.shape[0]} bounce rays, per ray: {per_nodes:.1f} nodes, {per_pk:.1f} packets")
    for cells, dbins in ((1, 1), (8, 1), (16, 2), (32, 4), (32, 8), (64, 8), (64, 16)):
        cell = np.clip(((o2 - lo) / (hi - lo) * cells).astype(np.int64), 0, cells - 1)
        # direction bin: octahedral-ish: sign bits + quantised |d| components
        a = np.abs(d2); q = np.clip((a / a.sum(axis=1, keepdims=True) * dbins).astype(np.int64), 0, dbins - 1)
        dkey = ((d2[:, 0] < 0) * 4 + (d2[:, 1] < 0) * 2 + (d2[:, 2] < 0)).astype(np.int64) * dbins * dbins + q[:, 0] * dbins + q[:, 1]
        key = ((cell[:, 0] * cells + cell[:, 1]) * cells + cell[:, 2]) * (8 * dbins * dbins) + dkey
        order = np.argsort(key, kind="stable")
        un, up, cnt = 0, 0, 0
        for s in range(0, len(order) - 63, 64):
            idx = order[s:s + 64]
            nodes = set(); leaves = {}
            for i in idx:
                nodes.update(vis[i][0])
                for l, k in vis[i][1]: leaves[l] = k
            un += len(nodes); up += sum(leaves.values()); cnt += 1
        print(f"cells {cells}^3 x dir bins {8*dbins*dbins}: union per 64-ray packet {un/cnt:.1f} nodes ({un/cnt/per_nodes:.1f}x), {up/cnt:.1f} packets ({up/cnt/per_pk:.1f}x)")

main()
