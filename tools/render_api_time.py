#!/usr/bin/env python3
"""Wall time of the drop-in render() (async worker thread, tile callbacks, host image) for the C2 frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minipath_amd as mp
ctx = mp.Context(0)
scene = mp.Scene(mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx))
st = mp.RenderSettings(64, 256, (1920, 1080), seed=0x5EED)
for it in range(6):
    n = [0]
    t0 = time.perf_counter()
    cbs = (lambda b: None, lambda b, s: n.__setitem__(0, n[0] + 1)) if it < 3 else (None, None)
    prog = mp.render(scene, mp.Camera.teapot_view(), st, *cbs)
    prog.wait()
    t1 = time.perf_counter()
    img = prog.image()
    t2 = time.perf_counter()
    print(f"render(): {(t1-t0)*1e3:.1f} ms wall, {n[0]} finished callbacks, image copy {(t2-t1)*1e3:.1f} ms, elapsed() {prog.elapsed()*1e3:.1f} ms")
