#!/usr/bin/env python3
"""One mp_trace_rays call on incoherent (bounce-like) rays of the stand-in, for rocprofv3 --pmc passes (tools/prof_trace.sh reads
the LAST trace_rays_kernel dispatch).  usage: prof_trace.py [n_rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minipath_amd import scenes
from minipath_amd.scene import Context, TriangleBvh

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dev = torch.device("cuda:0")
bvh = TriangleBvh.build(*scenes.atrium(1, 1.0), ctx=Context(0))
g = torch.Generator(device=dev); g.manual_seed(1)
lo = torch.tensor([-17.0, 0.5, -10.0], device=dev); hi = torch.tensor([17.0, 13.0, 10.0], device=dev)
o = lo + (hi - lo) * torch.rand((n, 3), device=dev, generator=g)
d = torch.nn.functional.normalize(torch.randn((n, 3), device=dev, generator=g), dim=1)
h = bvh.intersect(o, d)
hit = h["prim"] != -1
o2 = (o + d * h["t"][:, None])[hit]
d2 = torch.nn.functional.normalize(torch.randn((o2.shape[0], 3), device=dev, generator=g), dim=1)
o2 = (o2 + 1e-3 * d2).contiguous()
torch.cuda.synchronize()
bvh.intersect(o2, d2)
torch.cuda.synchronize()
print("rays", o2.shape[0])
