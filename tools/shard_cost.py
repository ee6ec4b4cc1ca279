#!/usr/bin/env python3
"""How even are the N-GPU tile shards?  Renders the metric's frame once on one GPU, reads the per-tile shader-cycle counters
(mp_launch_extras.tile_cost) and prints max / mean of the per-rank cost sums for the static r::N partition and for the
longest-processing-time partition of bench.py --balance lpt, N = 2, 4, 8.  usage: shard_cost.py [spp] [depth] [packet_samples_in_flight]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import minipath_amd as mp
from minipath_amd import scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = mp.Context(0)
if len(sys.argv) > 3:
    ctx.set_option("packet_samples_in_flight", int(sys.argv[3]))
scene = mp.Scene(mp.TriangleBvh.build(*scenes.atrium(1, 1.0), ctx))
st = mp.RenderSettings(64, spp, (1920, 1080), seed=0x5EED, max_depth=depth)
fr = mp.FrameRenderer(scene, scenes.atrium_camera(), st)
fr.render(); torch.cuda.synchronize()
fr.tile_cost.zero_()
fr.render(); torch.cuda.synchronize()
cost = fr.tile_cost[: len(fr.tiles)].cpu().numpy().astype(np.float64)
print("tiles", len(cost), "cost min/mean/max", cost.min(), cost.mean(), cost.max())
for n in (2, 4, 8):
    static = np.array([cost[r::n].sum() for r in range(n)])
    loads = np.zeros(n)
    for i in sorted(range(len(cost)), key=lambda i: (-cost[i], i)):
        r = int(np.argmin(loads)); loads[r] += cost[i]
    print(f"N={n}: static max/mean {static.max() / static.mean():.4f}   lpt max/mean {loads.max() / loads.mean():.4f}")

# what one rank's launch costs against the ideal 1/N of the full frame (launch tails, fewer tiles per XCD queue)
def timed(fr_, reps=5):
    fr_.render(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fr_.render()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
full = timed(fr)
tiles = list(fr.tiles)
for n in (2, 4, 8):
    ms = [timed(mp.FrameRenderer(scene, scenes.atrium_camera(), st, tiles=tiles[r::n])) for r in range(n)]
    print(f"N={n}: full frame {full:.2f} ms, ideal {full / n:.2f} ms, slowest static shard {max(ms):.2f} ms (mean {sum(ms) / n:.2f}) -> {full / max(ms):.2f}x before the gather")
