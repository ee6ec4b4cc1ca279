#!/usr/bin/env python3
"""Wall time of the drop-in render() (async worker thread(s), tile callbacks, host image) against the bare device launch.
usage: render_api_time.py [teapot|atrium] [contexts] [render_batch_tiles]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
from minipath_amd import scenes

which = sys.argv[1] if len(sys.argv) > 1 else "teapot"
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctxs = [mp.Context(0) for _ in range(nctx)]
if len(sys.argv) > 3:
    for c in ctxs: c.set_option("render_batch_tiles", int(sys.argv[3]))
if which == "atrium":
    mesh = scenes.atrium(1, 1.0)
    scs = [mp.Scene(mp.TriangleBvh.build(*mesh, c)) for c in ctxs]
    cam = scenes.atrium_camera()
else:
    obj = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj")
    scs = [mp.Scene(mp.TriangleBvh.with_obj(obj, c)) for c in ctxs]
    cam = mp.Camera.teapot_view()
st = mp.RenderSettings(64, 256, (1920, 1080), seed=0x5EED)
fr = mp.FrameRenderer(scs[0], cam, st)
fr.render(); fr.untile(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): fr.render(); fr.untile(reuse=True)
torch.cuda.synchronize()
print(f"{which}: bare launch + un-tile {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per frame")
for it in range(8):
    n = [0]
    t0 = time.perf_counter()
    cbs = (lambda b: None, lambda b, s: n.__setitem__(0, n[0] + 1)) if it < 3 else (None, None)
    st_ = st if it < 6 else mp.RenderSettings(64, 256, (1920, 1080), seed=0x5EED, image_u8_only=True)   # the last two: the reference's u8 image only
    prog = mp.render_multi(scs, cam, st_, *cbs) if nctx > 1 else mp.render(scs[0], cam, st_, *cbs)
    prog.wait()
    t1 = time.perf_counter()
    img = prog.image()
    t2 = time.perf_counter()
    print(f"render() over {nctx} context(s){' (u8 image only)' if it >= 6 else ''}: {(t1-t0)*1e3:.1f} ms wall, {n[0]} finished callbacks, image copy {(t2-t1)*1e3:.1f} ms, elapsed() {prog.elapsed()*1e3:.1f} ms")
