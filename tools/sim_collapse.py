#!/usr/bin/env python3
"""Round-3 study: collapse thin inner nodes of the reference BVH8 into full 8-wide DEVICE nodes and count what a
walk of the collapsed tree costs, before any kernel work.

A child p of node N may be absorbed into N (its children take p's place in N's child list, order kept) when every child box of p
is contained, in floating point, in p's own box: IEEE subtraction / multiplication are monotone, so for a ray with finite inverse
direction the slab interval of a child g lies inside the slab interval of p, and g's two tests (t1_g <= hi_g, t1_g <= best.t at
g's pop) imply p's (t1_p <= hi_p, t1_p <= best.t at p's earlier pop: best.t only shrinks).  Children are still pushed ascending
and popped descending, so the DFS order of the leaves is unchanged.

The walk below is a numpy model (f32, unfused triangle test: counts only, not a parity check) of (a) one ray per walk = the
8-lane-group kernel's unit of work and (b) 64 camera rays per walk = the packet kernel.  Diagnostics only.

usage: sim_collapse.py [atrium|teapot] [detail] [n_rays]
       sim_collapse.py --packet-studies [atrium|teapot]   (interval rejection / two-slab exit of the packet walk's child tests)
"""
import sys, os, time
import numpy as np
sys.path.insert(0, "/root/repo")
import minipath_amd as mp
from minipath_amd import scenes

NULL = 0xFFFFFFF8
F = np.float32
C65535 = F(1.0) / F(65535.0)


def fma32(a, b, c):
    return (np.float64(a) * np.float64(b) + np.float64(c)).astype(np.float32)


def load(scene, detail):
    if scene == "teapot":
        host = mp.TriangleBvh.with_obj("/root/repo/tests/golden/teapot.obj")
    else:
        host = mp.TriangleBvh.build(*scenes.atrium(1, detail))
    i = host.info()
    inner, packets, shading, vn, vt = host.export()
    return inner, packets, i.root_link, np.array(list(i.bbox_min), F), np.array(list(i.bbox_max), F)


class RefTree:
    """reference tree with the decompressed box chain: per inner node the 8 child boxes and links"""

    def __init__(self, inner, packets, root, bmin, bmax):
        n = inner.shape[0]
        q = inner[:, :96].copy().view(np.uint16).reshape(n, 2, 3, 8)   # [node][min/max][axis][child]
        self.links = inner[:, 96:].copy().view(np.uint32).reshape(n, 8)
        self.cbox = np.zeros((n, 8, 6), F)
        self.nbox = np.zeros((n, 6), F)
        self.root = root
        self.rootbox = np.concatenate([bmin, bmax])
        pk = packets.copy().view(np.uint16).reshape(-1, 3, 3, 8)       # [packet][vertex][coord][lane]
        self.leaf = {}                                                 # first packet -> (v0, e1, e2) [m,3]
        stack = [(root, self.rootbox)]
        while stack:
            link, box = stack.pop()
            idx, cnt = link >> 3, link & 7
            mn, size = box[:3], (box[3:] - box[:3]).astype(F)
            if cnt == 0:
                self.nbox[idx] = box
                rel = q[idx].astype(F) * C65535                        # [2][3][8]
                cb = fma32(size[None, :, None], rel, mn[None, :, None])  # [2][3][8]
                self.cbox[idx] = np.concatenate([cb[0].T, cb[1].T], axis=1)
                for c in range(8):
                    if self.links[idx, c] != NULL:
                        stack.append((int(self.links[idx, c]), self.cbox[idx, c].copy()))
            else:
                v = pk[idx:idx + cnt].astype(F) * C65535               # [cnt][3][3][8]
                p = fma32(size[None, None, :, None], v, mn[None, None, :, None])
                p = p.transpose(0, 3, 1, 2).reshape(cnt * 8, 3, 3)     # [tri][vertex][coord]
                real = np.any(pk[idx:idx + cnt].transpose(0, 3, 1, 2).reshape(cnt * 8, 9) != 0, axis=1)
                m = int(np.nonzero(real)[0].max()) + 1 if real.any() else 1
                p = p[:m]
                self.leaf[idx] = (p[:, 0], (p[:, 1] - p[:, 0]).astype(F), (p[:, 2] - p[:, 0]).astype(F))

    def children(self, idx):
        return [(int(self.links[idx, c]), self.cbox[idx, c]) for c in range(8) if self.links[idx, c] != NULL]

    def absorbable(self, link, box):
        """inner node whose child boxes are all ordered and FP-contained in its own box"""
        if link & 7:
            return False
        ch = self.children(link >> 3)
        for _, b in ch:
            if not (np.all(b[:3] <= b[3:]) and np.all(b[:3] >= box[:3]) and np.all(b[3:] <= box[3:])):
                return False
        return True


def area(b):
    s = np.maximum(b[3:] - b[:3], 0)
    return 2 * (s[0] * (s[1] + s[2]) + s[1] * s[2])


def build_device(ref, policy="area", limit=8):
    """returns nodes = list of (boxes[k,6], links[k]) ; link >= 0 : device node index ; link < 0 : leaf, first packet = -1-link"""
    nodes = []
    absorbed = [0]

    def make(idx):
        slots = ref.children(idx)
        if policy != "none":
            while True:
                cand = []
                for i, (l, b) in enumerate(slots):
                    if (l & 7) == 0 and ref.absorbable(l, b):
                        k = len(ref.children(l >> 3))
                        if len(slots) - 1 + k <= limit:
                            cand.append((i, k, area(b)))
                if not cand:
                    break
                if policy == "area":
                    i = max(cand, key=lambda c: c[2])[0]
                elif policy == "small":
                    i = min(cand, key=lambda c: (c[1], -c[2]))[0]
                else:
                    i = cand[0][0]
                l, b = slots[i]
                slots[i:i + 1] = ref.children(l >> 3)
                absorbed[0] += 1
        me = len(nodes)
        nodes.append(None)
        boxes = np.array([b for _, b in slots], F).reshape(-1, 6)
        links = []
        for l, b in slots:
            links.append(make(l >> 3) if (l & 7) == 0 else -1 - (l >> 3))
        nodes[me] = (boxes, np.array(links, np.int64))
        return me

    sys.setrecursionlimit(10000)
    root = make(ref.root >> 3)
    return nodes, root, absorbed[0]


def walk(nodes, root, ref, o, d, cnt):
    """P rays (rows of o, d) share one walk; per-ray decisions.  cnt: dict of counters."""
    P = o.shape[0]
    with np.errstate(divide="ignore"):
        inv = np.where(d == 0, F(np.inf), F(1) / d).astype(F)
    best = np.full(P, np.finfo(F).max, F)
    stack = [(root, np.ones(P, bool), None)]
    while stack:
        link, mask, box = stack.pop()
        cnt["pops"] += 1
        if box is not None:
            t1, _ = slab(box[None, :], o, inv, best)
            mask = mask & ~(t1[:, 0] > best)
        if not mask.any():
            cnt["culls"] += 1
            continue
        if link >= 0:
            boxes, links = nodes[link]
            cnt["nodes"] += 1
            cnt["boxes"] += len(links)
            cnt["hist"][len(links)] += 1
            t1, t2 = slab(boxes, o, inv, best)
            ok = (t1 <= t2) & mask[:, None]
            for c in range(len(links)):
                if ok[:, c].any():
                    stack.append((int(links[c]), ok[:, c].copy(), boxes[c]))
                    cnt["pushes"] += 1
        else:
            v0, e1, e2 = ref.leaf[-1 - link]
            cnt["leaves"] += 1
            cnt["tris"] += v0.shape[0]
            cnt["packets"] += (v0.shape[0] + 7) // 8
            t = mt(v0, e1, e2, o, d)                                   # [P, m]
            t = np.where(mask[:, None], t, np.inf)
            best = np.minimum(best, t.min(axis=1).astype(F))
    return best


def slab(boxes, o, inv, limit):
    a = (boxes[None, :, :3] - o[:, None, :]) * inv[:, None, :]
    c = (boxes[None, :, 3:] - o[:, None, :]) * inv[:, None, :]
    a = np.where(np.isnan(a), -np.inf, a)
    c = np.where(np.isnan(c), np.inf, c)
    lo, hi = np.minimum(a, c), np.maximum(a, c)
    t1 = np.maximum(np.maximum(lo[..., 0], 0), np.maximum(lo[..., 1], lo[..., 2]))
    t2 = np.minimum(np.minimum(hi[..., 0], limit[:, None]), np.minimum(hi[..., 1], hi[..., 2]))
    return t1.astype(F), t2.astype(F)


def mt(v0, e1, e2, o, d):
    dd = d[:, None, :]
    h = np.cross(dd, e2[None])
    det = (e1[None] * h).sum(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / det
        s = o[:, None, :] - v0[None]
        u = inv * (s * h).sum(-1)
        q = np.cross(s, e1[None])
        v = inv * (dd * q).sum(-1)
        t = inv * (e2[None] * q).sum(-1)
    ok = (u >= 0) & (v >= 0) & (u + v <= 1) & (t >= 0)
    return np.where(ok, t, np.inf)


def bounce_rays(ref, nodes, root, n, rng):
    lo = np.array([-17.0, 0.5, -10.0]); hi = np.array([17.0, 13.0, 10.0])
    if ref.rootbox[3] - ref.rootbox[0] < 20:   # teapot
        lo, hi = ref.rootbox[:3] - 1, ref.rootbox[3:] + 1
    o = (lo + (hi - lo) * rng.random((n * 2, 3))).astype(F)
    d = rng.standard_normal((n * 2, 3)).astype(F)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out_o, out_d = [], []
    dummy = new_cnt()
    for k in range(n * 2):
        t = walk(nodes, root, ref, o[k:k + 1], d[k:k + 1], dummy)[0]
        if t < 1e30:
            p = o[k] + d[k] * t
            d2 = rng.standard_normal(3).astype(F); d2 /= np.linalg.norm(d2)
            out_o.append((p + 1e-3 * d2).astype(F)); out_d.append(d2)
            if len(out_o) == n:
                break
    return np.array(out_o, F), np.array(out_d, F)


def camera_packets(scene, npk, rng, w=1920, h=1080):
    """npk packets of 64 camera rays: 2x2 pixels x 16 samples (the packet kernel's pass), via the oracle's sample_ray"""
    from oracle import pyoracle as po
    import ctypes as C
    if scene == "teapot":
        cam = po.teapot_camera()
    else:
        cam = po.Camera(); po.lib().mpo_camera_default(C.byref(cam))
        eye, at, fnum = scenes.ATRIUM_VIEW
        po.lib().mpo_camera_look_at(C.byref(cam), po.vec3(*eye), po.vec3(*at), po.vec3(0, 1, 0)); cam.f_number = fnum
    s = po.build_sampler(cam, w, h)
    out = []
    for _ in range(npk):
        x0, y0 = 2 * int(rng.integers(0, w // 2)), 2 * int(rng.integers(0, h // 2))
        o = np.zeros((64, 3), F); d = np.zeros((64, 3), F)
        for l in range(64):
            pix, sub = l // 16, l % 16
            r = po.sample_ray(s, x0 + pix % 2, y0 + pix // 2, int(rng.integers(0, 1 << 40)) + sub)
            o[l] = list(r.o); d[l] = list(r.d)
        out.append((o, d))
    return out



def packet_studies(scene="atrium", detail=1.0, npk=150):
    """Round-3 studies on the child-box tests of the packet walk (64 camera rays per walk, wide tree): (a) how many of the children a
    packet does NOT push could a conservative test on the packet's bounds (interval arithmetic on min / max of origins and inverse
    directions: monotone IEEE operations, exact) reject; (b) for how many does no ray survive two of the three slabs."""
    ref = RefTree(*load(scene, detail))
    rng = np.random.default_rng(1)
    pk = camera_packets(scene, npk, rng)
    nodes, root, _ = build_device(ref, "area", 8)
    tot = dict(visits=0, boxes=0, pushes=0, nopush=0, interval_reject=0, xy=0, xz=0, yz=0)

    def two(lo, hi, lim, mask, a, b):
        t1 = np.maximum(np.maximum(lo[..., a], 0), lo[..., b]); t2 = np.minimum(np.minimum(hi[..., a], lim[:, None]), hi[..., b])
        return (t1 <= t2) & mask[:, None]

    for o, d in pk:
        P = 64
        with np.errstate(divide="ignore"):
            inv = np.where(d == 0, F(np.inf), F(1) / d).astype(F)
        sg = np.sign(inv)
        uniform = all((sg[:, k] == sg[0, k]).all() for k in range(3)) and np.isfinite(inv).all()
        omin, omax, imin, imax = o.min(0), o.max(0), inv.min(0), inv.max(0)
        best = np.full(P, np.finfo(F).max, F)
        stack = [(root, np.ones(P, bool), None)]
        while stack:
            link, mask, box = stack.pop()
            if box is not None:
                t1, _ = slab(box[None, :], o, inv, best); mask = mask & ~(t1[:, 0] > best)
            if not mask.any():
                continue
            if link >= 0:
                boxes, links = nodes[link]
                tot["visits"] += 1; tot["boxes"] += len(links)
                with np.errstate(invalid="ignore"):
                    a = (boxes[None, :, :3] - o[:, None, :]) * inv[:, None, :]; c = (boxes[None, :, 3:] - o[:, None, :]) * inv[:, None, :]
                lo, hi = np.minimum(a, c), np.maximum(a, c)
                t1, t2 = slab(boxes, o, inv, best)
                ok = (t1 <= t2) & mask[:, None]
                rxy, rxz, ryz = two(lo, hi, best, mask, 0, 1), two(lo, hi, best, mask, 0, 2), two(lo, hi, best, mask, 1, 2)
                for cc in range(len(links)):
                    if ok[:, cc].any():
                        stack.append((int(links[cc]), ok[:, cc].copy(), boxes[cc])); tot["pushes"] += 1
                        continue
                    tot["nopush"] += 1
                    tot["xy"] += not rxy[:, cc].any(); tot["xz"] += not rxz[:, cc].any(); tot["yz"] += not ryz[:, cc].any()
                    if uniform:
                        b = boxes[cc]; L = []; U = []
                        for k in range(3):
                            pos = sg[0, k] > 0
                            near, far = (b[k], b[3 + k]) if pos else (b[3 + k], b[k])
                            a_lo, a_hi, b_lo, b_hi = F(near - omax[k]), F(near - omin[k]), F(far - omax[k]), F(far - omin[k])
                            if pos:
                                L.append(min(F(a_lo * imin[k]), F(a_lo * imax[k]))); U.append(max(F(b_hi * imin[k]), F(b_hi * imax[k])))
                            else:
                                L.append(min(F(a_hi * imin[k]), F(a_hi * imax[k]))); U.append(max(F(b_lo * imin[k]), F(b_lo * imax[k])))
                        tot["interval_reject"] += max(max(L), 0) > min(U)
            else:
                v0, e1, e2 = ref.leaf[-1 - link]
                t = mt(v0, e1, e2, o, d); t = np.where(mask[:, None], t, np.inf)
                best = np.minimum(best, t.min(axis=1).astype(F))
    print("per packet:", {k: round(v / len(pk), 2) for k, v in tot.items()})


def new_cnt():
    return {"pops": 0, "culls": 0, "nodes": 0, "boxes": 0, "pushes": 0, "leaves": 0, "tris": 0, "packets": 0, "hist": np.zeros(65, np.int64)}


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "atrium"
    detail = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    t0 = time.time()
    ref = RefTree(*load(scene, detail))
    print(f"{scene}: {ref.links.shape[0]} inner nodes, {len(ref.leaf)} leaves  ({time.time() - t0:.1f}s)")
    nch = np.array([len(ref.children(i)) for i in range(ref.links.shape[0])])
    print("children per node:", np.bincount(nch, minlength=9))
    ab = sum(1 for i in range(ref.links.shape[0]) for l, b in ref.children(i) if (l & 7) == 0 and ref.absorbable(l, b))
    inner_children = sum(1 for i in range(ref.links.shape[0]) for l, b in ref.children(i) if (l & 7) == 0)
    print(f"inner nodes whose children are all FP-nested in their own box: {ab} of {inner_children}")
    rng = np.random.default_rng(1)
    trees = {}
    for pol in ("none", "area", "small"):
        nodes, root, absorbed = build_device(ref, pol)
        k = np.array([len(l) for _, l in nodes])
        trees[pol] = (nodes, root)
        print(f"policy {pol:5s}: {len(nodes)} device nodes ({absorbed} absorbed), slots per node {np.bincount(k, minlength=9)}, mean {k.mean():.2f}")
    bo, bd = bounce_rays(ref, *trees["none"], n, rng)
    print(f"--- {bo.shape[0]} incoherent (bounce-like) rays, one ray per walk")
    base = None
    for pol, (nodes, root) in trees.items():
        c = new_cnt()
        ts = [walk(nodes, root, ref, bo[k:k + 1], bd[k:k + 1], c)[0] for k in range(bo.shape[0])]
        if base is None:
            base = ts
        assert np.array_equal(np.array(ts), np.array(base)), "collapsed walk changed a hit distance"
        m = bo.shape[0]
        steps = c["nodes"] + c["culls"] + c["leaves"]
        print(f"  {pol:5s}: per ray: pops {c['pops']/m:.2f} node steps {c['nodes']/m:.2f} culled pops {c['culls']/m:.2f} leaves {c['leaves']/m:.2f} "
              f"packets {c['packets']/m:.2f} boxes {c['boxes']/m:.1f} pushes {c['pushes']/m:.2f} | group-walk model VALU/ray "
              f"{(c['nodes']*55 + c['culls']*15 + c['leaves']*15 + c['packets']*68)/m:.0f}  visited-node slots {c['hist']}")
    pk = camera_packets(scene, max(8, n // 16), rng)
    print(f"--- {len(pk)} camera packets (2x2 pixels x 16 samples), 64 rays per walk")
    base = None
    for pol, (nodes, root) in trees.items():
        c = new_cnt()
        ts = np.array([walk(nodes, root, ref, o, d, c) for o, d in pk])
        if base is None:
            base = ts
        assert np.array_equal(ts, base), "collapsed walk changed a hit distance"
        m = len(pk)
        print(f"  {pol:5s}: per packet: pops {c['pops']/m:.2f} node visits {c['nodes']/m:.2f} culled pops {c['culls']/m:.2f} leaves {c['leaves']/m:.2f} "
              f"tris {c['tris']/m:.1f} boxes {c['boxes']/m:.1f} pushes {c['pushes']/m:.2f} | packet-walk model VALU/pass "
              f"{(c['boxes']*17 + c['tris']*20 + c['pops']*12 + c['nodes']*10 + c['leaves']*10)/m:.0f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--packet-studies":
        packet_studies(*(sys.argv[2:3] or ["atrium"]))
    else:
        main()
