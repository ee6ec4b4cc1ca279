#!/usr/bin/env python3
"""Throughput of mp_trace_rays (8-lane-group traversal) on incoherent rays: diffuse-bounce-like rays that start on the atrium's
surfaces (first hits of random interior rays) and leave in uniformly random directions.  Diagnostics only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minipath_amd import scenes
from minipath_amd.scene import Context, TriangleBvh

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    detail = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    dev = torch.device("cuda:0")
    bvh = TriangleBvh.build(*scenes.atrium(1, detail), ctx=Context(0))
    g = torch.Generator(device=dev); g.manual_seed(1)
    lo = torch.tensor([-17.0, 0.5, -10.0], device=dev); hi = torch.tensor([17.0, 13.0, 10.0], device=dev)
    o = lo + (hi - lo) * torch.rand((n, 3), device=dev, generator=g)
    d = torch.nn.functional.normalize(torch.randn((n, 3), device=dev, generator=g), dim=1)
    h = bvh.intersect(o, d)
    t = h["t"]; hit = h["prim"] != 0xFFFFFFFF if h["prim"].dtype != torch.int32 else h["prim"] != -1
    o2 = (o + d * t[:, None])[hit]
    d2 = torch.nn.functional.normalize(torch.randn((o2.shape[0], 3), device=dev, generator=g), dim=1)
    o2 = o2 + 1e-3 * d2
    print("bounce rays:", o2.shape[0], "of", n)
    for name, (oo, dd) in {"interior-random": (o, d), "surface-bounce": (o2.contiguous(), d2.contiguous())}.items():
        bvh.intersect(oo, dd); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): bvh.intersect(oo, dd)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{name}: {oo.shape[0] / dt / 1e6:.0f} Mrays/s ({dt * 1e3:.1f} ms)")

if __name__ == "__main__":
    main()
