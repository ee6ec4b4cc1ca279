#!/usr/bin/env python3
"""Static tile->rank patterns evaluated on measured per-tile costs (shader cycles from mp_launch_extras): max/mean rank load."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import minipath_amd as mp

def costs(scene_name):
    ctx = mp.Context(0)
    if scene_name == "atrium":
        from minipath_amd import scenes
        scene = mp.Scene(mp.TriangleBvh.build(*scenes.atrium(1, 1.0), ctx)); cam = scenes.atrium_camera(); spp = 64
    else:
        scene = mp.Scene(mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx))
        cam = mp.Camera.teapot_view(); spp = 256
    st = mp.RenderSettings(64, spp, (1920, 1080), seed=0x5EED)
    fr = mp.FrameRenderer(scene, cam, st)
    fr.render(); torch.cuda.synchronize()
    return fr.tiles, fr.tile_cost.cpu().numpy().astype(np.float64)[: len(fr.tiles)]

for name in ("teapot", "atrium"):
    tiles, c = costs(name)
    tx = np.array([t.min_x // 64 for t in tiles]); ty = np.array([t.min_y // 64 for t in tiles]); idx = np.arange(len(tiles))
    ncols = tx.max() + 1
    print(name, "tiles", len(tiles), "cost max/mean per tile", round(c.max() / c.mean(), 2))
    for N in (2, 4, 8):
        res = {}
        res["r::N (row-major)"] = idx % N
        for k in (1, 2, 3, 5, 7):
            res[f"(tx+{k}*ty)%N"] = (tx + k * ty) % N
        snake = np.where(ty % 2 == 0, tx, ncols - 1 - tx) + ty * ncols
        res["snake order %N"] = snake % N
        # LPT with true costs (lower bound for any static assignment)
        order = np.argsort(-c); load = np.zeros(N); lpt = np.zeros(len(c), int)
        for i in order:
            r = int(np.argmin(load)); lpt[i] = r; load[r] += c[i]
        res["LPT on measured cost"] = lpt
        out = []
        for k, a in res.items():
            loads = np.array([c[a == r].sum() for r in range(N)])
            out.append(f"{k}: {loads.max() / loads.mean():.3f}")
        print(f"  N={N}: " + " | ".join(out))
