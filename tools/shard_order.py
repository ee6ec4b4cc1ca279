#!/usr/bin/env python3
"""Does handing out the expensive tiles first shorten a small launch?  1/8 shard of the teapot / atrium frame, tile list in
row-major order vs sorted by measured per-tile cost (descending).  Diagnostics only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
from minipath_amd.distributed import plan_shards

def timeit(fr, n=3):
    fr.render(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fr.render()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

def main():
    scene_name = sys.argv[1] if len(sys.argv) > 1 else "teapot"
    ctx = mp.Context(0)
    if scene_name == "atrium":
        from minipath_amd import scenes
        scene = mp.Scene(mp.TriangleBvh.build(*scenes.atrium(1, 1.0), ctx)); cam = scenes.atrium_camera(); spp = 64
    else:
        scene = mp.Scene(mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx))
        cam = mp.Camera.teapot_view(); spp = 256
    st = mp.RenderSettings(64, spp, (1920, 1080), seed=0x5EED)
    full = mp.FrameRenderer(scene, cam, st)
    for world in (1, 8):
        tiles = list(plan_shards(full.tiles, world).shards[min(3, world - 1)])
        cost = [timeit(mp.FrameRenderer(scene, cam, st, tiles=[t]), 1) for t in tiles]
        base = timeit(mp.FrameRenderer(scene, cam, st, tiles=tiles))
        order = sorted(range(len(tiles)), key=lambda i: -cost[i])
        lpt = timeit(mp.FrameRenderer(scene, cam, st, tiles=[tiles[i] for i in order]))
        rev = timeit(mp.FrameRenderer(scene, cam, st, tiles=[tiles[i] for i in order[::-1]]))
        print(f"{scene_name} world {world}: row-major {base:.3f} ms, heavy-first {lpt:.3f} ms, light-first {rev:.3f} ms")

main()
