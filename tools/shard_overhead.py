#!/usr/bin/env python3
"""Event-bracketed time of mp_render_tiles_device for one rank's 1/8 shard (teapot 1080p x256); run under
rocprofv3 --kernel-trace --stats to compare with the kernel's own duration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
from minipath_amd.distributed import plan_shards

ctx = mp.Context(0)
scene = mp.Scene(mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx))
cam = mp.Camera.teapot_view()
st = mp.RenderSettings(64, 256, (1920, 1080), seed=0x5EED)
full = mp.FrameRenderer(scene, cam, st)
plan = plan_shards(full.tiles, 8)
fr = mp.FrameRenderer(scene, cam, st, tiles=plan.shards[3])
fr.render(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): fr.render()
b.record(); torch.cuda.synchronize()
print(f"shard launch, events: {a.elapsed_time(b) / 10:.3f} ms")
