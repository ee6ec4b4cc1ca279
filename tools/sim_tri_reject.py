#!/usr/bin/env python3
"""Round-3 study: how many of the triangle tests of a camera packet (64 rays of one or two pixels, wide tree) could a conservative
test on bounds of the packet's rays skip?  Interval evaluation of the Moeller-Trumbore expression sequence of triangle.rs:183-217
(every operation is monotone in each argument, so evaluating the same f32 operation at the corners of the operand intervals bounds
every ray's result: exact, no error analysis), on the origin / direction box of the packet widened by half its extent -- the bounds
a work unit's mask cache would hold.  numpy model (f32, fma through f64): counts only.

usage: sim_tri_reject.py [atrium|teapot] [packets] [pixels per packet: 1|2|4]
"""
import sys
import numpy as np
sys.path.insert(0, "/root/repo/tools")
sys.path.insert(0, "/root/repo")
from sim_collapse import RefTree, load, build_device, slab, F, fma32
from minipath_amd import scenes


def camera_packets(scene, npk, rng, pixels, w=1920, h=1080):
    from oracle import pyoracle as po
    import ctypes as C
    if scene == "teapot":
        cam = po.teapot_camera()
    else:
        cam = po.Camera(); po.lib().mpo_camera_default(C.byref(cam))
        eye, at, fnum = scenes.ATRIUM_VIEW
        po.lib().mpo_camera_look_at(C.byref(cam), po.vec3(*eye), po.vec3(*at), po.vec3(0, 1, 0)); cam.f_number = fnum
    s = po.build_sampler(cam, w, h)
    per = 64 // pixels
    out = []
    for _ in range(npk):
        x0, y0 = 2 * int(rng.integers(0, w // 2)), 2 * int(rng.integers(0, h // 2))
        o = np.zeros((64, 3), F); d = np.zeros((64, 3), F)
        for l in range(64):
            pix, sub = l // per, l % per
            r = po.sample_ray(s, x0 + pix % 2, y0 + pix // 2, int(rng.integers(0, 1 << 40)) + sub)
            o[l] = list(r.o); d[l] = list(r.d)
        out.append((o, d))
    return out


class Iv:
    """interval of f32 arrays"""
    def __init__(self, lo, hi):
        self.lo, self.hi = np.asarray(lo, F), np.asarray(hi, F)


def mul(a, b):
    c = [F(x) * F(y) for x in (a.lo, a.hi) for y in (b.lo, b.hi)]
    return Iv(np.minimum.reduce(c), np.maximum.reduce(c))


def fma(a, b, c):
    lo = [fma32(x, y, c.lo) for x in (a.lo, a.hi) for y in (b.lo, b.hi)]
    hi = [fma32(x, y, c.hi) for x in (a.lo, a.hi) for y in (b.lo, b.hi)]
    return Iv(np.minimum.reduce(lo), np.maximum.reduce(hi))


def neg(a):
    return Iv(-a.hi, -a.lo)


def sub(a, b):
    return Iv(a.lo - b.hi, a.hi - b.lo)


def dot(a, b):   # fma(az, bz, fma(ay, by, ax * bx))
    return fma(a[2], b[2], fma(a[1], b[1], mul(a[0], b[0])))


def fms(a, b, c):
    return fma(a, b, neg(c))


def interval_reject(v0, e1, e2, olo, ohi, dlo, dhi):
    """[m] bool: no ray with origin / direction inside the boxes can pass u >= 0, v >= 0, u + v <= 1, t >= 0"""
    m = v0.shape[0]
    P = lambda x: Iv(x, x)
    d = [Iv(np.full(m, dlo[k], F), np.full(m, dhi[k], F)) for k in range(3)]
    o = [Iv(np.full(m, olo[k], F), np.full(m, ohi[k], F)) for k in range(3)]
    E1 = [P(e1[:, k]) for k in range(3)]; E2 = [P(e2[:, k]) for k in range(3)]; V0 = [P(v0[:, k]) for k in range(3)]
    h = [fms(d[1], E2[2], mul(d[2], E2[1])), fms(d[2], E2[0], mul(d[0], E2[2])), fms(d[0], E2[1], mul(d[1], E2[0]))]
    det = dot(E1, h)
    straddle = (det.lo <= 0) & (det.hi >= 0)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = Iv(F(1) / det.hi, F(1) / det.lo)
        s = [sub(o[k], V0[k]) for k in range(3)]
        u = mul(inv, dot(s, h))
        q = [fms(s[1], E1[2], mul(s[2], E1[1])), fms(s[2], E1[0], mul(s[0], E1[2])), fms(s[0], E1[1], mul(s[1], E1[0]))]
        v = mul(inv, dot(d, q))
        t = mul(inv, dot(E2, q))
        rej = (u.hi < 0) | (v.hi < 0) | ((u.lo + v.lo) > 1) | (t.hi < 0)
    return rej & ~straddle & np.isfinite(inv.lo) & np.isfinite(inv.hi)


def mt_valid(v0, e1, e2, o, d):
    """[P, m] bool / t: the product's test without the best.t limit (f32, fma through f64)"""
    dd = [d[:, None, k] for k in range(3)]; oo = [o[:, None, k] for k in range(3)]
    E1 = [e1[None, :, k] for k in range(3)]; E2 = [e2[None, :, k] for k in range(3)]; V0 = [v0[None, :, k] for k in range(3)]
    fd = lambda a, b: fma32(a[2], b[2], fma32(a[1], b[1], F(a[0]) * F(b[0])))
    fm = lambda a, b, c: fma32(a, b, -c)
    h = [fm(dd[1], E2[2], dd[2] * E2[1]), fm(dd[2], E2[0], dd[0] * E2[2]), fm(dd[0], E2[1], dd[1] * E2[0])]
    det = fd(E1, h)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = F(1) / det
        s = [oo[k] - V0[k] for k in range(3)]
        u = inv * fd(s, h)
        q = [fm(s[1], E1[2], s[2] * E1[1]), fm(s[2], E1[0], s[0] * E1[2]), fm(s[0], E1[1], s[1] * E1[0])]
        v = inv * fd(dd, q)
        t = inv * fd(E2, q)
        ok = (u >= 0) & (v >= 0) & (u + v <= 1) & (t >= 0)
    return ok, np.where(ok, t, np.inf).astype(F)


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "atrium"
    npk = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    pixels = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    ref = RefTree(*load(scene, 1.0))
    nodes, root, _ = build_device(ref, "area", 8)
    rng = np.random.default_rng(3)
    pk = camera_packets(scene, npk, rng, pixels)
    tot = dict(leaves=0, packets=0, tris=0, hit_by_some=0, rejected=0, unsound=0, packets_all_rejected=0, pads={})
    for pad in (0.0, 0.5, 1.0):
        tot["pads"][pad] = 0
    for o, d in pk:
        with np.errstate(divide="ignore"):
            inv = np.where(d == 0, F(np.inf), F(1) / d).astype(F)
        best = np.full(64, np.finfo(F).max, F)
        stack = [(root, np.ones(64, bool), None)]
        bounds = {}
        for pad in tot["pads"]:
            eo, ed = (o.max(0) - o.min(0)) * F(pad), (d.max(0) - d.min(0)) * F(pad)
            bounds[pad] = ((o.min(0) - eo).astype(F), (o.max(0) + eo).astype(F), (d.min(0) - ed).astype(F), (d.max(0) + ed).astype(F))
        while stack:
            link, mask, box = stack.pop()
            if box is not None:
                t1, _ = slab(box[None, :], o, inv, best); mask = mask & ~(t1[:, 0] > best)
            if not mask.any():
                continue
            if link >= 0:
                boxes, links = nodes[link]
                t1, t2 = slab(boxes, o, inv, best)
                ok = (t1 <= t2) & mask[:, None]
                for cc in range(len(links)):
                    if ok[:, cc].any():
                        stack.append((int(links[cc]), ok[:, cc].copy(), boxes[cc]))
            else:
                v0, e1, e2 = ref.leaf[-1 - link]
                m = v0.shape[0]
                valid, t = mt_valid(v0, e1, e2, o, d)
                valid &= mask[:, None]
                some = valid.any(axis=0)
                tot["leaves"] += 1; tot["packets"] += (m + 7) // 8; tot["tris"] += m; tot["hit_by_some"] += int(some.sum())
                for pad, b in bounds.items():
                    rej = interval_reject(v0, e1, e2, *b)
                    tot["pads"][pad] += int(rej.sum())
                    tot["unsound"] += int((rej & some).sum())
                    if pad == 0.5:
                        tot["packets_all_rejected"] += sum(1 for p in range(0, m, 8) if rej[p:p + 8].all())
                t = np.where(valid, t, np.inf)
                best = np.minimum(best, t.min(axis=1).astype(F))
    n = len(pk)
    print(f"{scene}, {n} packets of {pixels} pixel(s) x {64 // pixels} samples; per packet:")
    print(f"  leaves {tot['leaves'] / n:.2f}  leaf packets {tot['packets'] / n:.2f}  triangle tests {tot['tris'] / n:.1f}  hit by some ray {tot['hit_by_some'] / n:.2f}")
    for pad, r in tot["pads"].items():
        print(f"  bounds widened by {pad} extents: {r / n:.1f} rejected ({100.0 * r / tot['tris']:.1f} %)")
    print(f"  leaf packets with every triangle rejected (0.5): {tot['packets_all_rejected'] / n:.2f}   rejected-but-hit (must be 0): {tot['unsound']}")


if __name__ == "__main__":
    main()
