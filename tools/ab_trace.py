#!/usr/bin/env python3
"""mp_trace_rays throughput on incoherent rays for several builds of the library (each in its own process).  usage: ab_trace.py lib.so ..."""
import os, subprocess, sys
code = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from minipath_amd import scenes
from minipath_amd.scene import Context, TriangleBvh
ctx = Context(0); dev = torch.device("cuda:0")
bvh = TriangleBvh.build(*scenes.atrium(1, 1.0), ctx=ctx)
n = 8_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
lo = torch.tensor([-17.0, 0.5, -10.0], device=dev); hi = torch.tensor([17.0, 13.0, 10.0], device=dev)
o = lo + (hi - lo) * torch.rand((n, 3), device=dev, generator=g)
d = torch.nn.functional.normalize(torch.randn((n, 3), device=dev, generator=g), dim=1)
h = bvh.intersect(o, d); hit = h["prim"] != -1
o2 = (o + d * h["t"][:, None])[hit]
d2 = torch.nn.functional.normalize(torch.randn((o2.shape[0], 3), device=dev, generator=g), dim=1)
o2 = (o2 + 1e-3 * d2).contiguous()
bvh.intersect(o2, d2); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): bvh.intersect(o2, d2)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"{o2.shape[0] / dt / 1e6:.0f} Mrays/s ({dt * 1e3:.2f} ms)")
'''
for so in [""] + sys.argv[1:]:
    env = dict(os.environ, MINIPATH_HIP_SO=so)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    print((so or "default").ljust(36), out[-1] if out else "failed", flush=True)
