#!/usr/bin/env python3
"""Would sorting incoherent (bounce) rays into 64-ray packets pay?  Bounce-like rays leaving the atrium's surfaces are keyed by
(origin grid cell, direction bin), sorted, cut into packets of 64; for each packet the union of inner nodes / leaf packets its rays
visit (what the packet walk would pay) is compared with the per-ray average (what a single ray needs).
`sim_sorted_packets.py window ...`: the same for bounce rays that start inside one screen window.  Diagnostics only."""
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

def main_window():
    """`sim_sorted_packets.py window [detail] [win] [spp]`: direction-sorting bounce rays that start inside one screen window (nearby
    origins): union of visited nodes / packets per 64-ray packet vs the per-ray average."""
    sys.argv.pop(1)
    detail = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
    win = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    b = po.Bvh.build(*scenes.atrium(1, detail))
    cam = po.Camera(); po.lib().mpo_camera_default(C.byref(cam)); po.lib().mpo_camera_look_at(C.byref(cam), po.vec3(-16.0, 4.2, 0.8), po.vec3(12.0, 5.5, -0.5), po.vec3(0, 1, 0)); cam.f_number = 4.0
    s = po.build_sampler(cam, 1920, 1080)
    L = po.lib(); rng = np.random.default_rng(5)
    for (x0, y0) in ((900, 500), (300, 800), (1500, 300)):
        o, d = [], []
        for y in range(y0, y0 + win):
            for x in range(x0, x0 + win):
                for smp in range(spp):
                    r = po.sample_ray(s, x, y, L.mpo_sample_key(1, 1920, spp, x, y, smp))
                    o.append([r.o[0], r.o[1], r.o[2]]); d.append([r.d[0], r.d[1], r.d[2]])
        o = np.array(o, np.float32); d = np.array(d, np.float32)
        t, prim, u, v = b.trace(o, d)
        hit = prim != 0xFFFFFFFF
        o2 = (o + d * t[:, None])[hit]
        d2 = rng.normal(size=o2.shape).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
        o2 = (o2 + 1e-3 * d2).astype(np.float32)
        vis = visits(b, o2, d2)
        per_nodes = np.mean([len(a) for a, _ in vis]); per_pk = np.mean([sum(k for _, k in l) for _, l in vis])
        print(f"window ({x0},{y0}) {win}x{win}x{spp}: {o2.shape[0]} bounce rays, per ray {per_nodes:.1f} nodes {per_pk:.1f} packets")
        for dbins in (0, 1, 2, 4, 8, 16):
            if dbins == 0:
                order = np.arange(o2.shape[0])
            else:
                a = np.abs(d2); q = np.clip((a / a.sum(axis=1, keepdims=True) * dbins).astype(np.int64), 0, dbins - 1)
                key = ((d2[:, 0] < 0) * 4 + (d2[:, 1] < 0) * 2 + (d2[:, 2] < 0)).astype(np.int64) * dbins * dbins + q[:, 0] * dbins + q[:, 1]
                order = np.argsort(key, kind="stable")
            un = up = cnt = 0
            for st in range(0, len(order) - 63, 64):
                nodes = set(); leaves = {}
                for i in order[st:st + 64]:
                    nodes.update(vis[i][0])
                    for l, k in vis[i][1]: leaves[l] = k
                un += len(nodes); up += sum(leaves.values()); cnt += 1
            print(f"   dir bins {8*dbins*dbins:5d}: union per packet {un/cnt:.1f} nodes ({un/cnt/per_nodes:.1f}x) {up/cnt:.1f} packets ({up/cnt/per_pk:.1f}x)")

if __name__ == "__main__":
    main_window() if len(sys.argv) > 1 and sys.argv[1] == "window" else main()
