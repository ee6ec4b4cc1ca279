#!/usr/bin/env python3
"""Frame time of the metric's workload against the samples of a pixel in flight per pass (ctx option packet_samples_in_flight).
usage: s_sweep.py [atrium|teapot] [spp] [packet_mask_cache: 0|1|2]   Diagnostics only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
from minipath_amd import scenes

which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = mp.Context(0)
if len(sys.argv) > 3:
    ctx.set_option("packet_mask_cache", int(sys.argv[3]))
if which == "atrium":
    scene, cam = mp.Scene(mp.TriangleBvh.build(*scenes.atrium(1, 1.0), ctx)), scenes.atrium_camera()
else:
    scene, cam = mp.Scene(mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx)), mp.Camera.teapot_view()
fr = mp.FrameRenderer(scene, cam, mp.RenderSettings(64, spp, (1920, 1080), seed=0x5EED))
for S in (4, 8, 16, 32, 64):
    ctx.set_option("packet_samples_in_flight", S)
    fr.render(); torch.cuda.synchronize()
    fr.rebalance()
    t0 = time.perf_counter()
    for _ in range(3): fr.render()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{which} S={S}: {dt * 1e3:.2f} ms  {1920 * 1080 * spp / dt / 1e9:.2f} Grays/s", flush=True)
