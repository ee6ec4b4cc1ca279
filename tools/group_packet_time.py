#!/usr/bin/env python3
"""camera rays of an object group on the packet walk.  Times the teapot frame (1080p x 64 spp, reference
semantics) as a plain TriangleBvh, as a group of two teapots side by side on the packet kernel, and the same group on the
8-lane-group kernel (what groups ran on before round 3).  Diagnostics only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import minipath_amd as mp

ctx = mp.Context(0)
teapot = mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx)
two = mp.ObjectGroup([teapot, teapot], np.array([[-3.4, 0, 0], [3.4, 0, 0]], np.float32))
cam = mp.Camera.default().look_at((0, 2, 16), (0, 1.5, 0), (0, 1, 0)).f_number(4.8).focus_distance(16.0)
for name, obj, trav in (("one teapot, packets", teapot, "packets"), ("two-teapot group, packets", two, "packets"), ("two-teapot group, 8-lane groups", two, "groups")):
    fr = mp.FrameRenderer(mp.Scene(obj), cam, mp.RenderSettings(64, 64, (1920, 1080), seed=0x5EED, traversal=trav))
    fr.render(); torch.cuda.synchronize(); fr.rebalance()
    t0 = time.perf_counter()
    for _ in range(5): fr.render()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {dt * 1e3:.2f} ms  {1920 * 1080 * 64 / dt / 1e9:.2f} Grays/s", flush=True)
