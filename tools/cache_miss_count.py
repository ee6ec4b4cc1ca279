#!/usr/bin/env python3
"""Slow-path calls of the packet walk's mask cache per work unit and pass (a build with -DMP_PROF_MISSES: tools/build_variant.sh prof
"-DMP_PROF_MISSES", run with MINIPATH_HIP_SO=variants/libmp_prof.so).  usage: cache_miss_count.py [atrium|teapot] [spp]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
from minipath_amd import scenes, _lib

which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = mp.Context(0)
if which == "atrium":
    scene, cam = mp.Scene(mp.TriangleBvh.build(*scenes.atrium(1, 1.0), ctx)), scenes.atrium_camera()
else:
    scene, cam = mp.Scene(mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx)), mp.Camera.teapot_view()
fr = mp.FrameRenderer(scene, cam, mp.RenderSettings(64, spp, (1920, 1080), seed=0x5EED))
fr.render(); torch.cuda.synchronize()
lib = _lib.lib()
out = (C.c_ulonglong * 4)()
lib.mp_prof_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert lib.mp_prof_read(out, 1) == 0
fr.render(); torch.cuda.synchronize()
assert lib.mp_prof_read(out, 0) == 0
S = 16 if spp >= 64 else 8 if spp >= 32 else 4
units = 1920 * 1080 // (64 // S)
passes = units * (spp // S)
print(f"{which} x{spp}: per unit: node-mask slow paths {out[0] / units:.1f}, leaf-mask slow paths {out[1] / units:.1f}, bounds (re)set {out[3] / units:.2f};"
      f" per pass: node visits {out[2] / passes:.1f}")
