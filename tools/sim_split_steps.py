#!/usr/bin/env python3
"""Follow-up to sim_policies.py: 4 / 2 lanes per ray with SPLIT steps -- a node visit takes ceil(children / lanes) node steps (so
16 or 32 rays of a wave each test up to 4 or 2 children per step), a packet 8 / lanes leaf steps -- replayed on oracle op traces
of bounce rays in the stand-in with the real child count of every visited node.  Prints the child-count histograms and VALU per
ray.  Diagnostics only (uses the oracle: test infrastructure)."""
import ctypes as C, sys, os, time
import numpy as np
sys.path.insert(0, "/root/repo")
from oracle import pyoracle as po
import minipath_amd as mp
from minipath_amd import scenes

def get(n=4096, detail=1.0):
    host = mp.TriangleBvh.build(*scenes.atrium(1, detail))
    i = host.info()
    inner, packets, shading, vn, vt, mat = host.export(with_material=True)
    links = inner.view(np.uint32).reshape(-1, 32)[:, 24:]
    nchild = np.array([ (np.nonzero(l != 0xFFFFFFF8)[0].max() + 1) if np.any(l != 0xFFFFFFF8) else 0 for l in links])
    print("child count histogram (all nodes):", np.bincount(nchild, minlength=9))
    b = po.Bvh.from_arrays(inner, packets, shading, vn, vt, i.root_link, list(i.bbox_min), list(i.bbox_max), material=mat)
    rng = np.random.default_rng(1)
    lo = np.array([-17.0, 0.5, -10.0]); hi = np.array([17.0, 13.0, 10.0])
    o = (lo + (hi - lo) * rng.random((n*2, 3))).astype(np.float32)
    d = rng.standard_normal((n*2, 3)).astype(np.float32)
    t, prim, u, v = b.trace(o, d)
    hit = prim != 0xFFFFFFFF
    dn = d / np.linalg.norm(d, axis=1, keepdims=True)
    o2 = (o + dn * t[:, None])[hit][:n]
    d2 = rng.standard_normal((o2.shape[0], 3)).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o2 = (o2 + 1e-3 * d2).astype(np.float32)
    L = po.lib(); buf = (C.c_uint8 * 8192)(); lbuf = (C.c_uint32 * 8192)()
    out = []
    for k in range(o2.shape[0]):
        r = po.ray_new(o2[k], d2[k])
        m = L.mpo_bvh_intersect_ops(b.h, C.byref(r), buf, lbuf, 8192)
        ops = []
        for j in range(m):
            op = buf[j]
            if op == 1: ops.append(("N", int(nchild[lbuf[j] >> 3])))
            elif op >= 8: ops.append(("L", op - 8))
            else: ops.append(("C", 0))
        out.append(ops)
    return out

def sim(rays, lanes_per_ray, cA, cP, cB, cO, per_wave=64, policy="both"):
    """lanes_per_ray in (8, 4, 2): a node visit takes ceil(nchild/lanes) A-steps, a packet 8/lanes B-steps; nslots = 64/lanes."""
    ns = 64 // lanes_per_ray
    total = 0; iters = 0
    for w0 in range(0, len(rays) - per_wave + 1, per_wave):
        q = []
        for r in rays[w0:w0+per_wave]:
            steps = []
            for kind, v in r:
                if kind == "N": steps += ["A"] * max(1, -(-v // lanes_per_ray))
                elif kind == "C": steps.append("P")
                else: steps.append("P"); steps += ["B"] * (v * 8 // lanes_per_ray)
            q.append(steps)
        S = [None] * ns
        while True:
            for k in range(ns):
                if S[k] is None and q: S[k] = [q.pop(0), 0]
            if all(s is None for s in S): break
            iters += 1; c = cO
            didA = didP = didB = False
            # phase A: every slot whose next step is A or P takes it
            for s in S:
                if s is None: continue
                st = s[0][s[1]]
                if st == "A": didA = True; s[1] += 1
                elif st == "P":
                    didP = True; s[1] += 1
            # phase B: slots whose next step (possibly after a P) is B
            for s in S:
                if s is None or s[1] >= len(s[0]): continue
                if s[0][s[1]] == "B": didB = True; s[1] += 1
            c += (cA if didA else (cP if didP else 0)) + (cB if didB else 0)
            total += c
            for k in range(ns):
                if S[k] is not None and S[k][1] >= len(S[k][0]): S[k] = None
    n = (len(rays) // per_wave) * per_wave
    return round(total / n, 1), round(iters / (n / per_wave), 1)

if __name__ == "__main__":
    R = get()
    nn = [v for r in R for k, v in r if k == "N"]
    print("visited nodes: child count histogram", np.bincount(nn, minlength=9), "mean", np.mean(nn), "P(n>4)", np.mean(np.array(nn) > 4), "P(n>2)", np.mean(np.array(nn) > 2))
    for pw in (64, 256):
        print("per_wave", pw)
        print("  8 lanes/ray  ", sim(R, 8, 55, 15, 62, 10, pw))
        print("  4 lanes/ray  ", sim(R, 4, 55, 15, 62, 10, pw))
        print("  2 lanes/ray  ", sim(R, 2, 55, 15, 62, 10, pw))
