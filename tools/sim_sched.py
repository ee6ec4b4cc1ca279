#!/usr/bin/env python3
"""Scheduling study: replays real per-ray traversal op traces (from the oracle) through models of the wave-level
traversal loop (8 groups x 8 lanes, dynamic ray fetch) and counts VALU work per ray for alternative loop structures.
Diagnostics only (not product, not a test)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po

def traces(w=256, h=256, spp=2, blocks=200, seed=0x5EED):
    b = po.Bvh.from_obj("tests/golden/teapot.obj")
    s = po.build_sampler(po.teapot_camera(), w, h)
    L = po.lib()
    buf = (C.c_uint8 * 4096)()
    lbuf = (C.c_uint32 * 4096)()
    rng = np.random.default_rng(1)
    out = []  # list of wave-passes: each = list of 64 op lists
    for _ in range(blocks):
        bx, by = int(rng.integers(0, w // 8)), int(rng.integers(0, h // 8))
        for smp in range(spp):
            wave = []
            for p in range(64):
                x, y = bx * 8 + p % 8, by * 8 + p // 8
                r = po.sample_ray(s, x, y, L.mpo_sample_key(seed, w, spp, x, y, smp))
                n = L.mpo_bvh_intersect_ops(b.h, C.byref(r), buf, lbuf, 4096)
                wave.append(list(buf[:n]))
                LINKS.setdefault(id(wave), []).append([(buf[i], lbuf[i]) for i in range(n)])
            out.append(wave)
    return out

A_NODE, A_POP, B_PK, OVH = 45, 12, 52, 30
LINKS = {}

def sim_current(wave, prefilter=True):
    """one pop (node/leaf/culled) per group per iteration, then one packet per group per iteration"""
    rays = [r for r in wave if not (prefilter and r == [1])] if prefilter else list(wave)
    # prefilter: rays whose only op is the root node with no children pushed
    q = list(rays); groups = [None] * 8; cost = 0; iters = 0; useful = 0
    while True:
        for g in range(8):
            if groups[g] is None and q:
                groups[g] = {"ops": q.pop(0), "i": 0, "pk": 0}
        if all(g is None for g in groups): break
        iters += 1; c = OVH; didA = didPop = didB = False
        for g in groups:
            if g is None: continue
            if g["pk"] == 0 and g["i"] < len(g["ops"]):
                op = g["ops"][g["i"]]; g["i"] += 1
                if op == 1: didA = True; useful += A_NODE
                else: didPop = True
                if op >= 8: g["pk"] = op - 8
        for g in groups:
            if g is None: continue
            if g["pk"] > 0: g["pk"] -= 1; didB = True; useful += B_PK
        c += (A_NODE if didA else (A_POP if didPop else 0)) + (B_PK if didB else 0)
        cost += c
        for k in range(8):
            g = groups[k]
            if g is not None and g["pk"] == 0 and g["i"] >= len(g["ops"]): groups[k] = None
    return cost, iters, useful / 8.0

def sim_whilewhile(wave, prefilter=True):
    rays = [r for r in wave if not (prefilter and r == [1])] if prefilter else list(wave)
    q = list(rays); groups = [None] * 8; cost = 0; iters = 0
    while True:
        for g in range(8):
            if groups[g] is None and q:
                groups[g] = {"ops": q.pop(0), "i": 0, "pk": 0}
        if all(g is None for g in groups): break
        cost += OVH
        # phase A: until every active group has a leaf or is done
        while True:
            did = False; node = False
            for g in groups:
                if g is None: continue
                if g["pk"] == 0 and g["i"] < len(g["ops"]):
                    op = g["ops"][g["i"]]; g["i"] += 1; did = True
                    if op == 1: node = True
                    if op >= 8: g["pk"] = op - 8
            if not did: break
            cost += (A_NODE if node else A_POP) + 8; iters += 1
        while True:
            did = False
            for g in groups:
                if g is None: continue
                if g["pk"] > 0: g["pk"] -= 1; did = True
            if not did: break
            cost += B_PK + 6; iters += 1
        for k in range(8):
            g = groups[k]
            if g is not None and g["pk"] == 0 and g["i"] >= len(g["ops"]): groups[k] = None
    return cost, iters, 0

if __name__ == "__main__":
    W = traces()
    nr = sum(len(w) for w in W)
    trav = sum(1 for w in W for r in w if r != [1])
    ops = [op for w in W for r in w for op in r]
    print("rays", nr, "traversed", trav, "pops/ray", len(ops) / nr, "culled", ops.count(0) / nr, "nodes", ops.count(1) / nr,
          "leaves", sum(1 for o in ops if o >= 8) / nr, "packets", sum(o - 8 for o in ops if o >= 8) / nr)
    # ray-packet model: one wave = 64 rays, shared canonical DFS; cost per distinct (non-culled) node / packet visited by any ray
    cost = 0; ku = []; 
    for w in W:
        nodes, leaves = {}, {}
        for r in LINKS[id(w)]:
            for op, link in r:
                if op == 1: nodes[link] = nodes.get(link, 0) + 1
                elif op >= 8: leaves[link] = leaves.get(link, 0) + 1
        cost += len(nodes) * 190 + sum((l & 7) * (8 * 52 + 20) for l in leaves) + 40
        ku += [v for v in nodes.values()] 
    print(f"ray-packet   VALU/ray {cost / nr:7.1f}   mean active rays per visited node {np.mean(ku):.1f}")
    for name, f in (("current", sim_current), ("while-while", sim_whilewhile)):
        c = sum(f(w)[0] for w in W); it = sum(f(w)[1] for w in W); u = sum(f(w)[2] for w in W)
        print(f"{name:12s} VALU/ray {c / nr:7.1f}  VALU/traversed {c / trav:7.1f}  iters/wave-pass {it / len(W):6.1f}  useful-frac {u / c if c else 0:.2f}")
