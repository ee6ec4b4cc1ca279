#!/usr/bin/env python3
"""Differential test of mp_trace_rays (8-lane-group traversal) against the oracle on many random rays per scene, incl. zero / -0 /
axis-parallel direction components and origins inside the scene (the atrium scene exercises the wide device tree).  usage: fuzz_trace.py [rays_per_scene] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
from oracle import pyoracle as po
from tests import meshes

def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = mp.Context(0)
bad = 0
for si, name in enumerate(("soup_300", "grid_40", "sphere_24", "flat_plane", "soup_5000", "teapot", "atrium")):
    if name == "teapot":
        scene = mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx)
        ob = po.Bvh.from_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"))
    elif name == "atrium":  # the Sponza stand-in at a tenth of its detail: a tree whose thin top nodes the WIDE device tree absorbs
        from minipath_amd import scenes
        pos, nrm, tex, tri = scenes.atrium(1, 0.1)
        scene = mp.TriangleBvh.build(pos, nrm, tex, tri, ctx); ob = po.Bvh.build(pos, nrm, tex, tri)
        assert scene.device_tree()[3] > 0
    else:
        pos, nrm, tex, tri = meshes.make(name)
        scene = mp.TriangleBvh.build(pos, nrm, tex, tri, ctx); ob = po.Bvh.build(pos, nrm, tex, tri)
    bmin, bmax = scene.get_bounding_box()
    o, d = meshes.random_rays(n, seed + si, bmin, bmax)
    rng = np.random.default_rng(seed + 100 + si)
    k = n // 10
    o[:k] = (np.asarray(bmin) + rng.random((k, 3), dtype=np.float32) * (np.asarray(bmax) - np.asarray(bmin))).astype(np.float32)  # inside
    d[k:2 * k] = rng.normal(size=(k, 3)).astype(np.float32)                                                                      # any direction
    out = scene.intersect(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda())
    torch.cuda.synchronize()
    t, prim, u, v = ob.trace(o, d)
    gp = out["prim"].cpu().numpy().view(np.uint32)
    hit = prim != 0xFFFFFFFF
    m = int(np.sum(gp != prim)) + int(np.sum(bits(out["t"].cpu().numpy()) != bits(t))) + int(np.sum(bits(out["u"].cpu().numpy())[hit] != bits(u)[hit])) + int(np.sum(bits(out["v"].cpu().numpy())[hit] != bits(v)[hit]))
    print(f"{name}: {n} rays, {int(hit.sum())} hits, {m} differing values", flush=True)
    bad += m
sys.exit(1 if bad else 0)
