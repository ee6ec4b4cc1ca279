#!/usr/bin/env python3
"""Frame time of the metric's workload (stand-in 1080p, packets) against resident waves per SIMD (ctx option blocks_per_cu).
usage: occupancy_render.py [spp]   Diagnostics only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
from minipath_amd import scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = mp.Context(0)
scene = mp.Scene(mp.TriangleBvh.build(*scenes.atrium(1, 1.0), ctx))
fr = mp.FrameRenderer(scene, scenes.atrium_camera(), mp.RenderSettings(64, spp, (1920, 1080), seed=0x5EED))
for bpc in (8, 7, 6, 5, 4, 3, 2):
    ctx.set_option("blocks_per_cu", bpc)
    fr.render(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fr.render()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"waves/SIMD {bpc}: {dt * 1e3:.2f} ms  {1920 * 1080 * spp / dt / 1e9:.2f} Grays/s", flush=True)
