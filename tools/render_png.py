#!/usr/bin/env python3
"""Render a frame through render()/RenderProgress and write PNGs (visual sanity check).  usage: render_png.py OUTDIR"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minipath_amd as mp
from minipath_amd import io, scenes
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
os.makedirs(out, exist_ok=True)
ctx = mp.Context(0)
for name, scene, cam, res, spp in (
    ("teapot", mp.Scene(mp.TriangleBvh.with_obj("tests/golden/teapot.obj", ctx)), mp.Camera.teapot_view(), (640, 480), 64),
    ("atrium", mp.Scene(mp.TriangleBvh.build(*scenes.atrium(1, 1.0), ctx)), scenes.atrium_camera(), (640, 360), 32),
):
    t = time.time()
    rp = mp.render(scene, cam, mp.RenderSettings(64, spp, res))
    rp.wait()
    img = rp.image()
    img[..., 3] = 255  # opaque for viewing (alpha = hit fraction in the reference's output)
    io.save_png(os.path.join(out, name + ".png"), img)
    print(name, res, spp, "spp", "%.3f s" % (time.time() - t), "elapsed", rp.elapsed())
