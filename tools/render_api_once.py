#!/usr/bin/env python3
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minipath_amd as mp
ctx = mp.Context(0)
scene = mp.Scene(mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx))
st = mp.RenderSettings(64, 256, (1920, 1080), seed=0x5EED)
for it in range(2):
    prog = mp.render(scene, mp.Camera.teapot_view(), st, None, None); prog.wait()
    print("elapsed ms", prog.elapsed() * 1e3)
