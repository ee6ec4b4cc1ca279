#!/usr/bin/env python3
"""Scheduling study for the incoherent-ray tracer (round 2): replays per-ray traversal op traces of bounce-like rays in the
Sponza stand-in (from the oracle, over the product builder's arrays) through cost models of alternative wave-level loop
structures -- the shipped 8 groups x 8 lanes loop (node and leaf phase in every iteration), one phase per iteration chosen by
majority, a pool of ray slots decoupled from the lane groups, 4 / 2 / 1 lanes per ray, a leaf phase that only runs when k groups
wait -- and prints VALU wave-instructions per ray for each.  Costs per phase are the instruction counts of the shipped ISA.
Diagnostics only (uses the oracle: test infrastructure, not product).  Results: profiles/r02_notes.md."""
import ctypes as C, sys, os, time
import numpy as np
sys.path.insert(0, "/root/repo")
from oracle import pyoracle as po
import minipath_amd as mp
from minipath_amd import scenes

def get_traces(n=4096, detail=0.5):
    host = mp.TriangleBvh.build(*scenes.atrium(1, detail))
    i = host.info()
    inner, packets, shading, vn, vt, mat = host.export(with_material=True)
    b = po.Bvh.from_arrays(inner, packets, shading, vn, vt, i.root_link, list(i.bbox_min), list(i.bbox_max), material=mat)
    rng = np.random.default_rng(1)
    lo = np.array([-17.0, 0.5, -10.0]); hi = np.array([17.0, 13.0, 10.0])
    o = (lo + (hi - lo) * rng.random((n*2, 3))).astype(np.float32)
    d = rng.standard_normal((n*2, 3)).astype(np.float32)
    t, prim, u, v = b.trace(o, d)
    hit = prim != 0xFFFFFFFF
    dn = d / np.linalg.norm(d, axis=1, keepdims=True)
    o2 = (o + dn * t[:, None])[hit][:n]
    d2 = rng.standard_normal((o2.shape[0], 3)).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o2 = (o2 + 1e-3 * d2).astype(np.float32)
    L = po.lib()
    buf = (C.c_uint8 * 8192)(); lbuf = (C.c_uint32 * 8192)()
    out = []
    for k in range(o2.shape[0]):
        r = po.ray_new(o2[k], d2[k])
        m = L.mpo_bvh_intersect_ops(b.h, C.byref(r), buf, lbuf, 8192)
        out.append(list(buf[:m]))
    return out

def sim(rays, policy, cA=45, cP=15, cB=75, cO=8, cF=25, slots=8, th=4, both_cost=None):
    """rays: list of op lists.  A wave = 64 rays fed to `slots` ray slots (8 groups compute per iteration).
    policy: 'current' (A and B both every iteration), 'majority' (one phase per iteration, B when >= th groups wait or no A)"""
    total = 0; iters = 0; served = 0
    for w0 in range(0, len(rays) - 63, 64):
        q = [list(r) for r in rays[w0:w0+64]]
        S = [None] * slots
        while True:
            fin = False
            for k in range(slots):
                if S[k] is None and q:
                    S[k] = {"ops": q.pop(0), "i": 0, "pk": 0}; fin = True
            if all(s is None for s in S): break
            iters += 1
            c = cO + (cF if fin else 0)
            wantA = [s for s in S if s is not None and s["pk"] == 0 and s["i"] < len(s["ops"])]
            wantB = [s for s in S if s is not None and s["pk"] > 0]
            if policy == "current":
                node = False
                for s in wantA:
                    op = s["ops"][s["i"]]; s["i"] += 1
                    if op == 1: node = True
                    if op >= 8: s["pk"] = op - 8
                    served += 1
                for s in wantB: s["pk"] -= 1; served += 1
                c += (cA if node else (cP if wantA else 0)) + (cB if wantB else 0)
            else:
                doB = (len(wantB) >= th) or not wantA
                if doB:
                    for s in wantB[:8]: s["pk"] -= 1; served += 1
                    c += cB
                else:
                    node = False
                    for s in wantA[:8]:
                        op = s["ops"][s["i"]]; s["i"] += 1
                        if op == 1: node = True
                        if op >= 8: s["pk"] = op - 8
                        served += 1
                    c += cA if node else cP
            total += c
            for k in range(slots):
                s = S[k]
                if s is not None and s["pk"] == 0 and s["i"] >= len(s["ops"]): S[k] = None
    n = (len(rays) // 64) * 64
    return total / n, iters / (n / 64), served / max(iters, 1)

if __name__ == "__main__":
    t0 = time.time()
    R = get_traces()
    ops = [o for r in R for o in r]
    n = len(R)
    print("rays", n, "pops/ray", len(ops)/n, "culled", ops.count(0)/n, "nodes", ops.count(1)/n, "leaves", sum(1 for o in ops if o >= 8)/n,
          "packets", sum(o-8 for o in ops if o >= 8)/n, "time", time.time()-t0)
    print("current            ", sim(R, "current", cO=25, cF=0))
    print("current lean ovh   ", sim(R, "current"))
    for th in (2,3,4,5,6,8):
        print("majority th", th, sim(R, "majority", th=th))
    for slots in (12, 16, 24):
        for th in (4, 6, 8):
            print("slots", slots, "th", th, sim(R, "majority", th=th, slots=slots))

def sim_fixed(rays, nslots, cA, cP, cB, cO, cF, policy="both", th=0.5, per_wave=64):
    """nslots ray slots per wave, every slot has its own lanes (served every iteration if its phase runs)."""
    total = 0; iters = 0
    for w0 in range(0, len(rays) - per_wave + 1, per_wave):
        q = [list(r) for r in rays[w0:w0+per_wave]]
        S = [None] * nslots
        while True:
            fin = False
            for k in range(nslots):
                if S[k] is None and q:
                    S[k] = {"ops": q.pop(0), "i": 0, "pk": 0}; fin = True
            if all(s is None for s in S): break
            iters += 1
            c = cO + (cF if fin else 0)
            wantA = [s for s in S if s is not None and s["pk"] == 0 and s["i"] < len(s["ops"])]
            wantB = [s for s in S if s is not None and s["pk"] > 0]
            doA = doB = True
            if policy == "majority":
                doB = (len(wantB) >= th * nslots) or not wantA
                doA = not doB
            if doA and wantA:
                node = False
                for s in wantA:
                    op = s["ops"][s["i"]]; s["i"] += 1
                    if op == 1: node = True
                    if op >= 8: s["pk"] = op - 8
                c += cA if node else cP
            if doB and wantB:
                # in 'both' mode a slot that just popped a leaf link also gets its first packet this iteration (as the real loop does)
                for s in ([s for s in S if s is not None and s["pk"] > 0] if policy == "both" else wantB): s["pk"] -= 1
                c += cB
            elif policy == "both":
                nb = [s for s in S if s is not None and s["pk"] > 0]
                if nb:
                    for s in nb: s["pk"] -= 1
                    c += cB
            total += c
            for k in range(nslots):
                s = S[k]
                if s is not None and s["pk"] == 0 and s["i"] >= len(s["ops"]): S[k] = None
    n = (len(rays) // per_wave) * per_wave
    return round(total / n, 1), round(iters / (n / per_wave), 1)

if __name__ == "__main__":
    print("--- fixed slots")
    print("8x8 both      ", sim_fixed(R, 8, 45, 15, 75, 8, 25))
    print("8x8 majority.5", sim_fixed(R, 8, 45, 15, 75, 8, 25, "majority", 0.5))
    print("16x4 both     ", sim_fixed(R, 16, 95, 20, 135, 8, 25))
    for th in (0.25, 0.375, 0.5, 0.625):
        print("16x4 majority", th, sim_fixed(R, 16, 95, 20, 135, 8, 25, "majority", th))
    print("32x2 both     ", sim_fixed(R, 32, 175, 25, 260, 8, 25))
    for th in (0.25, 0.375, 0.5):
        print("32x2 majority", th, sim_fixed(R, 32, 175, 25, 260, 8, 25, "majority", th))
    for th in (0.25, 0.375, 0.5):
        print("64x1 majority", th, sim_fixed(R, 64, 330, 30, 500, 8, 25, "majority", th))
    print("64x1 majority 0.375 per_wave 256", sim_fixed(R, 64, 330, 30, 500, 8, 25, "majority", 0.375, per_wave=256))
    print("16x4 majority 0.375 per_wave 256", sim_fixed(R, 16, 95, 20, 135, 8, 25, "majority", 0.375, per_wave=256))
    print("8x8 both per_wave 256", sim_fixed(R, 8, 45, 15, 75, 8, 25, per_wave=256))
    print("--- pooled (decoupled) with realistic overheads")
    for slots in (12, 16, 24, 32):
        for th in (6, 8):
            print("pool slots", slots, "th", th, sim(R, "majority", cA=95, cP=60, cB=105, cO=0, cF=10, slots=slots, th=th))

def sim_bthresh(rays, k, cA=55, cP=15, cB=60, cO=8, cF=0, per_wave=64, nslots=8):
    total = 0; iters = 0; nB = 0
    for w0 in range(0, len(rays) - per_wave + 1, per_wave):
        q = [list(r) for r in rays[w0:w0+per_wave]]
        S = [None] * nslots
        while True:
            for j in range(nslots):
                if S[j] is None and q: S[j] = {"ops": q.pop(0), "i": 0, "pk": 0}
            if all(s is None for s in S): break
            iters += 1; c = cO
            wantA = [s for s in S if s is not None and s["pk"] == 0 and s["i"] < len(s["ops"])]
            node = False
            for s in wantA:
                op = s["ops"][s["i"]]; s["i"] += 1
                if op == 1: node = True
                if op >= 8: s["pk"] = op - 8
            if wantA: c += cA if node else cP
            wantB = [s for s in S if s is not None and s["pk"] > 0]
            # after A: groups still able to do A next iteration
            canA = [s for s in S if s is not None and s["pk"] == 0 and s["i"] < len(s["ops"])]
            if wantB and (len(wantB) >= k or not canA):
                for s in wantB: s["pk"] -= 1
                c += cB; nB += 1
            total += c
            for j in range(nslots):
                s = S[j]
                if s is not None and s["pk"] == 0 and s["i"] >= len(s["ops"]): S[j] = None
    n = (len(rays) // per_wave) * per_wave
    return round(total / n, 1), round(iters / (n / per_wave), 1), round(nB / (n / per_wave), 1)

if __name__ == "__main__":
    print("--- A always, B when >= k groups wait")
    for k in (1, 2, 3, 4, 5, 6):
        print("k", k, sim_bthresh(R, k))
