#!/usr/bin/env python3
"""Kernel time of a whole frame and of one rank's 1/8 shard for different numbers of samples in flight (work-unit sizes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
from minipath_amd.distributed import plan_shards

def timeit(fr):
    fr.render(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fr.render(); fr.render(); fr.render(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 3

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
    plan = plan_shards(full.tiles, 8)
    shard = mp.FrameRenderer(scene, cam, st, tiles=plan.shards[3])
    ref = None
    for S in (8, 16, 32, 64):
        ctx.set_option("packet_samples_in_flight", S)
        t_full, t_shard = timeit(full), timeit(shard)
        img = full.tile_buf.clone()
        ok = True if ref is None else bool(torch.equal(img.view(torch.int32), ref.view(torch.int32)))
        if ref is None: ref = img
        print(f"{scene_name} S={S}: full {t_full:.3f} ms, 1/8 shard {t_shard:.3f} ms (x8 = {8*t_shard:.2f}), same bits {ok}")

main()
