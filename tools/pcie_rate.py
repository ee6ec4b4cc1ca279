#!/usr/bin/env python3
"""PCIe-inclusive frame rate: render + un-tile + copy of the finished image to host memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
ctx = mp.Context(0)
scene = mp.Scene(mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx))
st = mp.RenderSettings(64, 256, (1920, 1080), seed=0x5EED)
fr = mp.FrameRenderer(scene, mp.Camera.teapot_view(), st)
h8 = torch.empty((1080, 1920, 4), dtype=torch.uint8).pin_memory()
h32 = torch.empty((1080, 1920, 4), dtype=torch.float32).pin_memory()
for name, want32 in (("u8 image (8.3 MB)", False), ("u8 + f32 images (41.5 MB)", True)):
    for it in range(4):
        if it == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        fr.render()
        img, img8 = fr.untile(reuse=True)
        h8.copy_(img8, non_blocking=True)
        if want32: h32.copy_(img, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {dt*1e3:.2f} ms per frame incl. device-to-host copy = {1920*1080*256/dt/1e9:.1f} Grays/s")
