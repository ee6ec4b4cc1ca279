#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
S = int(sys.argv[1])
ctx = mp.Context(0)
scene = mp.Scene(mp.TriangleBvh.with_obj(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "teapot.obj"), ctx))
st = mp.RenderSettings(64, 256, (1920, 1080), seed=0x5EED)
ctx.set_option("packet_samples_in_flight", S)
fr = mp.FrameRenderer(scene, mp.Camera.teapot_view(), st)
fr.render(); torch.cuda.synchronize()
