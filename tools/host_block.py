#!/usr/bin/env python3
"""Does mp_render_tiles_device_ex block the host while the stream is busy?  (pageable H2D copy of the tile list)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
ctx = mp.Context(0)
scene = mp.Scene(mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx))
st = mp.RenderSettings(64, 256, (1920, 1080), seed=0x5EED)
fr = mp.FrameRenderer(scene, mp.Camera.teapot_view(), st)
fr.render(); torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); fr.render(); t1 = time.perf_counter(); fr.render(); t2 = time.perf_counter(); fr.render(); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"host time of 3 back-to-back render() calls: {(t1-t0)*1e3:.3f} {(t2-t1)*1e3:.3f} {(t3-t2)*1e3:.3f} ms; drain {(t4-t3)*1e3:.3f} ms")
