#!/usr/bin/env python3
"""Frame time of the path extension against the fused kernel's form (ctx option paths_pooled: 0 = one pass per walk, 2 / 3 = two /
up to four passes pooled per walk).  usage: paths_mode_time.py [teapot|atrium] [spp] [depth]   Diagnostics only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
from minipath_amd import scenes

which = sys.argv[1] if len(sys.argv) > 1 else "teapot"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ctx = mp.Context(0)
if which == "atrium":
    scene, cam = mp.Scene(mp.TriangleBvh.build(*scenes.atrium(1, 1.0), ctx)), scenes.atrium_camera()
else:
    scene, cam = mp.Scene(mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx)), mp.Camera.teapot_view()
fr = mp.FrameRenderer(scene, cam, mp.RenderSettings(64, spp, (1920, 1080), seed=0x5EED, max_depth=depth))
for mode in (0, 2, 3):
    ctx.set_option("paths_pooled", mode)
    fr.render(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fr.render()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    seg = int(fr.segments.item())
    print(f"{which} x{spp} depth {depth}, paths_pooled={mode}: {dt * 1e3:.2f} ms  {seg / dt / 1e9:.2f} Grays/s ({seg / (1920 * 1080 * spp):.2f} segments per sample)", flush=True)
