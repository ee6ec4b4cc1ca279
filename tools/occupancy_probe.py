#!/usr/bin/env python3
"""mp_trace_rays throughput on incoherent rays as a function of resident waves per SIMD (ctx option blocks_per_cu): tells a
latency-bound kernel (throughput ~ proportional to occupancy) from an issue-bound one.  Diagnostics only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minipath_amd import scenes
from minipath_amd.scene import Context, TriangleBvh

ctx = Context(0)
dev = torch.device("cuda:0")
bvh = TriangleBvh.build(*scenes.atrium(1, 1.0), ctx=ctx)
n = 8_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
lo = torch.tensor([-17.0, 0.5, -10.0], device=dev); hi = torch.tensor([17.0, 13.0, 10.0], device=dev)
o = lo + (hi - lo) * torch.rand((n, 3), device=dev, generator=g)
d = torch.nn.functional.normalize(torch.randn((n, 3), device=dev, generator=g), dim=1)
h = bvh.intersect(o, d)
hit = h["prim"] != -1
o2 = (o + d * h["t"][:, None])[hit]
d2 = torch.nn.functional.normalize(torch.randn((o2.shape[0], 3), device=dev, generator=g), dim=1)
o2 = (o2 + 1e-3 * d2).contiguous()
for bpc in (8, 6, 4, 3, 2, 1):
    ctx.set_option("blocks_per_cu", bpc)
    bvh.intersect(o2, d2); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): bvh.intersect(o2, d2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"waves/SIMD {bpc}: {o2.shape[0] / dt / 1e6:.0f} Mrays/s ({dt * 1e3:.1f} ms)", flush=True)
