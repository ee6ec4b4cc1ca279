#!/usr/bin/env python3
"""Cost of splitting a frame's samples over progressive passes (MP_FLAG_ACCUMULATE): teapot 1080p x256 in 1, 4, 16 passes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
ctx = mp.Context(0)
scene = mp.Scene(mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx))
st = mp.RenderSettings(64, 256, (1920, 1080), seed=0x5EED)
fr = mp.FrameRenderer(scene, mp.Camera.teapot_view(), st)
fr.render(); torch.cuda.synchronize(); fr.rebalance()
ref = None
for passes in (1, 4, 16):
    per = 256 // passes
    for it in range(3):
        if it == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        nxt = 0
        for p in range(passes):
            nxt = fr.render_pass(nxt, per if p < passes - 1 else 0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    img = fr.tile_buf.clone()
    same = True if ref is None else bool(torch.equal(img.view(torch.int32), ref.view(torch.int32)))
    if ref is None: ref = img
    print(f"{passes:3d} passes of {per} spp: {dt*1e3:.2f} ms per frame, same bits {same}")
