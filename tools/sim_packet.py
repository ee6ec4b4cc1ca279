#!/usr/bin/env python3
"""Packet-coherence study: for the S=8 footprint (4x2 pixels x 8 samples) counts, per wave-pass, the union of nodes / leaf
triangles the packet walk visits vs the per-ray averages.  Diagnostics only."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po

def study(b, s, w, h, spp, blocks, S=8, BW=4, BH=2, seed=1):
    L = po.lib(); buf = (C.c_uint8 * 8192)(); lbuf = (C.c_uint32 * 8192)()
    rng = np.random.default_rng(2)
    tot_nodes = tot_leaf_pk = rays = 0; un_nodes = un_pk = 0; act = []
    for _ in range(blocks):
        bx, by = int(rng.integers(0, w // BW)), int(rng.integers(0, h // BH))
        nodes, leaves = {}, {}
        for p in range(BW * BH):
            x, y = bx * BW + p % BW, by * BH + p // BW
            for smp in range(S):
                r = po.sample_ray(s, x, y, L.mpo_sample_key(seed, w, spp, x, y, smp))
                n = L.mpo_bvh_intersect_ops(b.h, C.byref(r), buf, lbuf, 8192)
                rays += 1
                for i in range(n):
                    if buf[i] == 1: nodes[lbuf[i]] = nodes.get(lbuf[i], 0) + 1; tot_nodes += 1
                    elif buf[i] >= 8: leaves[lbuf[i]] = leaves.get(lbuf[i], 0) + 1; tot_leaf_pk += buf[i] - 8
        un_nodes += len(nodes); un_pk += sum(l & 7 for l in leaves); act += list(nodes.values())
    print(f"per ray: nodes {tot_nodes/rays:.1f} packets {tot_leaf_pk/rays:.1f} | per 64-ray packet: union nodes {un_nodes/blocks:.1f} union packets {un_pk/blocks:.1f} "
          f"| mean active rays per visited node {np.mean(act):.1f}")

if __name__ == "__main__":
    from minipath_amd import scenes
    which = sys.argv[1] if len(sys.argv) > 1 else "teapot"
    if which == "teapot":
        b = po.Bvh.from_obj("tests/golden/teapot.obj"); s = po.build_sampler(po.teapot_camera(), 1920, 1080)
    else:
        b = po.Bvh.build(*scenes.atrium(1, float(sys.argv[2]) if len(sys.argv) > 2 else 0.25))
        cam = po.Camera(); po.lib().mpo_camera_default(C.byref(cam)); po.lib().mpo_camera_look_at(C.byref(cam), po.vec3(-16.0,4.2,0.8), po.vec3(12.0,5.5,-0.5), po.vec3(0,1,0)); cam.f_number=4.0
        s = po.build_sampler(cam, 1920, 1080)
    study(b, s, 1920, 1080, 64, 300)
