#!/usr/bin/env python3
"""Cost of splitting a frame's samples over progressive passes (MP_FLAG_ACCUMULATE): teapot 1080p x256 in 1, 4, 16 passes;
`progressive_cost.py single`: the time of ONE pass of k = 1..64 samples per pixel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
ctx = mp.Context(0)
scene = mp.Scene(mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx))
st = mp.RenderSettings(64, 256, (1920, 1080), seed=0x5EED)
fr = mp.FrameRenderer(scene, mp.Camera.teapot_view(), st)
fr.render(); torch.cuda.synchronize(); fr.rebalance()
if len(sys.argv) > 1 and sys.argv[1] == "single":   # time of ONE progressive pass of k samples per pixel (sample_count 256)
    for k in (1, 2, 4, 8, 16, 32, 64):
        fr.render_pass(0, k); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(5): fr.render_pass(k, k)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"pass of {k:2d} spp: {dt*1e3:.3f} ms = {1920*1080*k/dt/1e9:.1f} Grays/s")
    sys.exit(0)
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
