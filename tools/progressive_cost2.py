#!/usr/bin/env python3
"""Time of ONE progressive pass of k samples per pixel (teapot 1080p, sample_count 256)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
ctx = mp.Context(0)
scene = mp.Scene(mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx))
st = mp.RenderSettings(64, 256, (1920, 1080), seed=0x5EED)
fr = mp.FrameRenderer(scene, mp.Camera.teapot_view(), st)
fr.render(); torch.cuda.synchronize(); fr.rebalance()
for k in (1, 2, 4, 8, 16, 32, 64):
    fr.render_pass(0, k); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(5): fr.render_pass(k, k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"pass of {k:2d} spp: {dt*1e3:.3f} ms = {1920*1080*k/dt/1e9:.1f} Grays/s")
