#!/usr/bin/env python3
"""Per-rank kernel time of the static tile partition, measured on ONE GPU (each rank's shard rendered alone): max/mean = the
load imbalance an N-GPU frame would see.  Diagnostics only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
from minipath_amd.distributed import plan_shards

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
    for world in (1, 2, 4, 8):
        plan = plan_shards(full.tiles, world)
        ms = []
        for r in range(world):
            fr = mp.FrameRenderer(scene, cam, st, tiles=plan.shards[r])
            fr.render(); torch.cuda.synchronize()
            if len(sys.argv) > 2: fr.rebalance()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fr.render(); fr.render(); fr.render(); b.record(); torch.cuda.synchronize()
            ms.append(a.elapsed_time(b) / 3)
        print(f"{scene_name} world {world}: max {max(ms):.3f} ms mean {sum(ms)/len(ms):.3f} ms imbalance {max(ms)/(sum(ms)/len(ms)):.3f}  ideal {ms and sum(ms)/world:.3f}")

main()
