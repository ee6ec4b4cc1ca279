"""The argument behind the packet walk's triangle masks (kernels.hip, tri_may_hit), checked on the CPU: the Moeller-Trumbore
expression sequence evaluated on intervals (corner evaluation of monotone f32 operations) never rejects a triangle that some ray
inside the bounds hits.  numpy model of tools/sim_tri_reject.py against per-ray f32 tests on camera packets of the teapot."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_interval_rejection_is_conservative():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sim_tri_reject as st
    from sim_collapse import RefTree, load

    ref = RefTree(*load("teapot", 1.0))
    rng = np.random.default_rng(5)
    packets = st.camera_packets("teapot", 8, rng, 2, w=192, h=108)
    rejected = hit = wrong = 0
    for o, d in packets:
        for pad in (0.0, 0.25):
            eo, ed = (o.max(0) - o.min(0)) * np.float32(pad), (d.max(0) - d.min(0)) * np.float32(pad)
            b = ((o.min(0) - eo).astype(np.float32), (o.max(0) + eo).astype(np.float32), (d.min(0) - ed).astype(np.float32), (d.max(0) + ed).astype(np.float32))
            for v0, e1, e2 in ref.leaf.values():
                rej = st.interval_reject(v0, e1, e2, *b)
                ok, _ = st.mt_valid(v0, e1, e2, o, d)
                some = ok.any(axis=0)
                rejected += int(rej.sum()); hit += int(some.sum()); wrong += int((rej & some).sum())
    assert wrong == 0
    assert hit > 0 and rejected > 10 * hit  # the test sees hits, and the bounds reject most of the rest


def test_interval_rejection_of_child_boxes_is_conservative():
    """... and the node masks (kernels.hip, bounds_may_hit): a child box the bounds reject is passed by no ray inside them."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sim_tri_reject as st
    from sim_collapse import RefTree, build_device, load, slab

    F = np.float32
    ref = RefTree(*load("teapot", 1.0))
    nodes, _, _ = build_device(ref, "area", 8)
    rng = np.random.default_rng(9)
    rejected = passed = wrong = 0
    for o, d in st.camera_packets("teapot", 8, rng, 2, w=192, h=108):
        with np.errstate(divide="ignore"):
            inv = np.where(d == 0, F(np.inf), F(1) / d).astype(F)
        if not np.isfinite(inv).all() or any((np.sign(inv[:, k]) != np.sign(inv[0, k])).any() for k in range(3)):
            continue  # the device takes the plain walk for such a pass
        sg = np.sign(inv[0])
        for pad in (0.0, 0.25):
            eo, ei = (o.max(0) - o.min(0)) * F(pad), (inv.max(0) - inv.min(0)) * F(pad)
            omin, omax = (o.min(0) - eo).astype(F), (o.max(0) + eo).astype(F)
            imin, imax = (inv.min(0) - ei).astype(F), (inv.max(0) + ei).astype(F)
            for k in range(3):  # an inverse-direction bound never crosses zero (mask_cache_begin_pass)
                if np.sign(imin[k]) != sg[k] or imin[k] == 0: imin[k] = inv[:, k].min()
                if np.sign(imax[k]) != sg[k] or imax[k] == 0: imax[k] = inv[:, k].max()
            for boxes, _links in nodes:
                t1, t2 = slab(boxes, o, inv, np.full(64, np.finfo(F).max, F))
                some = (t1 <= t2).any(axis=0)
                L, U = [], []
                for k in range(3):
                    y, z = (boxes[:, k] - omax[k]).astype(F), (boxes[:, 3 + k] - omin[k]).astype(F)
                    lo_src, hi_src = (y, z) if sg[k] > 0 else (z, y)
                    L.append(np.minimum(lo_src * imin[k], lo_src * imax[k]).astype(F))
                    U.append(np.maximum(hi_src * imin[k], hi_src * imax[k]).astype(F))
                T1 = np.maximum(np.maximum(L[0], F(0)), np.maximum(L[1], L[2])); T2 = np.minimum(U[0], np.minimum(U[1], U[2]))
                rej = T1 > T2
                rejected += int(rej.sum()); passed += int(some.sum()); wrong += int((rej & some).sum())
    assert wrong == 0
    assert passed > 0 and rejected > 0


def test_corner_evaluation_bounds_every_operation():
    """The one lemma both masks rest on: an f32 operation that is monotone in each operand (fl(a*b), fl(a*b+c), fl(a-b), fl(1/x) away from
    zero) takes its extremes over a box of operands at the box's corners -- so the same operation evaluated at the corners bounds its
    result for every operand inside.  Random boxes (spanning zero, tiny, huge, degenerate) and random points inside them, through the
    interval helpers of tools/sim_tri_reject.py."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sim_tri_reject as st
    from sim_collapse import fma32

    F = np.float32
    rng = np.random.default_rng(17)
    n = 20000

    def boxes():
        scale = F(10.0) ** rng.integers(-6, 7, n).astype(F)
        c = (rng.standard_normal(n).astype(F) * scale).astype(F)
        w = (np.abs(rng.standard_normal(n)).astype(F) * scale * F(10.0) ** rng.integers(-4, 1, n).astype(F)).astype(F)
        w[rng.random(n) < 0.1] = 0  # degenerate boxes
        lo, hi = (c - w).astype(F), (c + w).astype(F)
        x = (lo + (hi - lo) * rng.random(n).astype(F)).astype(F)
        return st.Iv(lo, hi), np.clip(x, lo, hi).astype(F)

    (A, a), (B, b), (Z, z) = boxes(), boxes(), boxes()
    m = st.mul(A, B)
    p = (a * b).astype(F)
    assert np.all((m.lo <= p) & (p <= m.hi))
    f = st.fma(A, B, Z)
    q = fma32(a, b, z)
    assert np.all((f.lo <= q) & (q <= f.hi))
    d = st.sub(A, B)
    r = (a - b).astype(F)
    assert np.all((d.lo <= r) & (r <= d.hi))
    nz = (A.lo > 0) | (A.hi < 0)  # reciprocal: only on boxes that stay away from zero
    with np.errstate(divide="ignore", over="ignore"):
        inv_lo, inv_hi, ia = (F(1) / A.hi), (F(1) / A.lo), (F(1) / a)
    assert nz.any() and np.all((inv_lo[nz] <= ia[nz]) & (ia[nz] <= inv_hi[nz]))


def test_interval_rejection_on_adversarial_triangles():
    """The triangle masks' argument on inputs a camera never produces: triangle soups at scales from 1e-5 to 1e5 (needles, near-degenerate
    and exactly degenerate triangles among them), ray bundles from inside, grazing along edges and planes, with zero direction
    components; bounds = the bundle's own box and a wider one.  Never a rejected triangle that some ray of the bundle hits."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sim_tri_reject as st

    F = np.float32
    rng = np.random.default_rng(23)
    rejected = hit = wrong = 0
    for case in range(120):
        scale = F(10.0) ** F(rng.integers(-5, 6))
        m = 48
        v0 = (rng.standard_normal((m, 3)) * scale).astype(F)
        e1 = (rng.standard_normal((m, 3)) * scale * F(10.0) ** rng.integers(-3, 1, (m, 1))).astype(F)
        e2 = (rng.standard_normal((m, 3)) * scale * F(10.0) ** rng.integers(-3, 1, (m, 1))).astype(F)
        e2[:4] = e1[:4] * F(2)          # exactly degenerate (collinear edges)
        e2[4:8] = 0                     # zero edge
        # a bundle of 64 rays: origins near a point, directions near a direction -- or aimed along a triangle's edge / inside its plane
        o0 = (rng.standard_normal(3) * scale * 3).astype(F)
        kind = case % 4
        if kind == 0:
            d0 = rng.standard_normal(3)
        elif kind == 1:
            d0 = (v0[10] + F(0.5) * e1[10]) - o0      # at a point of an edge
        elif kind == 2:
            d0 = e1[11] + F(1e-3) * e2[11]            # inside a triangle's plane
            o0 = (v0[11] - F(2) * d0).astype(F)
        else:
            d0 = np.array([1.0, 0.0, 0.0])            # axis-aligned: zero components
        spread_o, spread_d = F(10.0) ** F(rng.integers(-4, 0)) * scale, F(10.0) ** F(rng.integers(-5, -1))
        o = (o0[None, :] + rng.standard_normal((64, 3)) * spread_o).astype(F)
        d = d0[None, :] + rng.standard_normal((64, 3)) * spread_d * (0.0 if (kind == 3 and case % 8 == 3) else 1.0)
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F)
        ok, _ = st.mt_valid(v0, e1, e2, o, d)
        some = ok.any(axis=0)
        for pad in (0.0, 0.25):
            eo, ed = (o.max(0) - o.min(0)) * F(pad), (d.max(0) - d.min(0)) * F(pad)
            rej = st.interval_reject(v0, e1, e2, (o.min(0) - eo).astype(F), (o.max(0) + eo).astype(F), (d.min(0) - ed).astype(F), (d.max(0) + ed).astype(F))
            rejected += int(rej.sum()); hit += int(some.sum()); wrong += int((rej & some).sum())
    assert wrong == 0
    assert hit > 50 and rejected > 1000
