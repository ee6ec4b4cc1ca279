"""CPU tests of the product's device trees (minipath_amd/csrc/device_tree.cpp, exported through mp_scene_device_tree): the WIDE
tree -- thin nodes of the reference tree absorbed into their parents -- must reach the same leaves in the same order as the
literal reference tree (ray_bvh_intersection.rs:26-62), which is what makes the GPU walks' hits bit-identical.

Checked here without a GPU: structure (slot counts, leaf order, node accounting), the floating-point containment every absorbed
node has to satisfy, the stack bound, and -- with a numpy model of the reference's explicit-stack walk run on both trees --
that every ray visits the same leaf sequence with the same closest distance.  (The model's triangle test is plain numpy: it
decides nothing about parity with the reference, only whether the two trees are equivalent under one and the same test.)"""
import numpy as np
import pytest

import minipath_amd as mp
from minipath_amd import scenes
from tests import meshes
from tests.conftest import TEAPOT

NULL = 0xFFFFFFF8
F = np.float32


def _host(name):
    if name == "teapot":
        return mp.TriangleBvh.with_obj(TEAPOT)
    if name.startswith("atrium"):
        return mp.TriangleBvh.build(*scenes.atrium(1, float(name.split(":")[1])))
    pos, nrm, tex, tri = meshes.make(name)
    return mp.TriangleBvh.build(pos, nrm, tex, tri)


def _slots(nodes, n):
    """real slots of device node n: list of (box f32[6], link)"""
    out = []
    for i in range(8):
        link = int(nodes[n, i, 6])
        if link != NULL:
            out.append((nodes[n, i, :6].view(F), link))
    return out


def _leaf_order(nodes, root):
    """leaf links in the order a full DFS (children descending, as popped) reaches them, and per inner slot its signature"""
    seq, sig = [], {}
    if root == NULL:
        return seq, sig
    if root & 63:
        return [root], sig
    stack = [(root, None)]
    while stack:
        link, _ = stack.pop()
        if link & 63:
            seq.append(link)
            continue
        for box, l in _slots(nodes, link >> 6):   # pushed ascending, popped descending
            stack.append((l, box))
    return seq, sig


def _subtree_signatures(nodes, root):
    """{node index: (first leaf link in ascending-child order, number of leaves)}"""
    first, count = {}, {}
    order = []
    stack = [root >> 6]
    while stack:
        n = stack.pop()
        order.append(n)
        for _, l in _slots(nodes, n):
            if (l & 63) == 0:
                stack.append(l >> 6)
    for n in reversed(order):
        f, c = None, 0
        for _, l in _slots(nodes, n):
            if l & 63:
                lf, lc = l, 1
            else:
                lf, lc = first[l >> 6], count[l >> 6]
            f = lf if f is None else f
            c += lc
        first[n], count[n] = f, c
    return first, count


SCENES = ["teapot", "atrium:0.05", "atrium:0.1", "atrium:0.5", "soup_5000", "grid_40", "sphere_24", "two_clusters", "sliver_fan"]


@pytest.mark.parametrize("name", SCENES)
def test_wide_tree_structure_and_containment(name):
    host = _host(name)
    wide, wroot, wbound, absorbed = host.device_tree()
    lit, lroot, lbound, labs = host.device_tree(literal=True)
    info = host.info()
    assert labs == 0 and lit.shape[0] == info.inner_count
    assert wide.shape[0] + absorbed == lit.shape[0]
    if lit.shape[0] == 0:
        assert wroot == lroot
        return
    # literal tree: node n = reference node n, links re-encoded in place
    inner = host.export()[0]
    ref_links = inner[:, 96:].copy().view(np.uint32).reshape(-1, 8)
    assert np.array_equal(lit[:, :, 6] == NULL, ref_links == NULL)
    # slot counts and the loop bound n
    for tree in (wide, lit):
        real = tree[:, :, 6] != NULL
        last = np.where(real.any(1), 8 - np.argmax(real[:, ::-1], axis=1), 0)
        assert np.array_equal(tree[:, 0, 7], last)
    assert np.all((wide[:, :, 6] != NULL).sum(1) >= 2) or name == "flat_plane"
    # wide nodes hold their real children first, without gaps
    real = wide[:, :, 6] != NULL
    assert np.all(real[:, :-1] >= real[:, 1:])
    # same leaves in the same order
    assert _leaf_order(wide, wroot)[0] == _leaf_order(lit, lroot)[0]
    # every reference node that is no longer a node of its own has all its child boxes FP-contained in its own box
    wfirst, wcount = _subtree_signatures(wide, wroot)
    lfirst, lcount = _subtree_signatures(lit, lroot)
    kept = set()
    for n in range(wide.shape[0]):
        for box, l in _slots(wide, n):
            if (l & 63) == 0:
                kept.add((wfirst[l >> 6], wcount[l >> 6], box.tobytes()))
    gone = 0
    for n in range(lit.shape[0]):
        for box, l in _slots(lit, n):
            if l & 63:
                continue
            c = l >> 6
            if (lfirst[c], lcount[c], box.tobytes()) in kept:
                continue
            gone += 1
            assert np.all(box[:3] <= box[3:])
            for cb, _ in _slots(lit, c):
                assert np.all(cb[:3] <= cb[3:]) and np.all(box[:3] <= cb[:3]) and np.all(cb[3:] <= box[3:]), (n, c)
    assert gone == absorbed
    # the stand-in's binary top is what the wide tree is for
    if name.startswith("atrium"):
        assert absorbed > 0
        assert (wide[:, 0, 7] == 2).sum() < (lit[:, 0, 7] == 2).sum()
    assert wbound >= 1 and lbound >= 1 and info.stack_bound == 0  # host-only scene: no device stack


def _leaf_tris(host):
    """first packet -> (v0, e1, e2) of the leaf's triangles, decompressed like the upload does (float64 fma: exact product, one
    more rounding than fmaf in rare halfway cases -- irrelevant for an equivalence test that uses the same values on both trees)"""
    inner, packets, *_ = host.export()
    return inner, packets.copy().view(np.uint16).reshape(-1, 3, 3, 8)


def _walk(nodes, root, tris_of, o, d):
    """the reference's walk (ray_bvh_intersection.rs:26-62) on a device-format tree: returns (leaf sequence, best t, max stack)"""
    with np.errstate(divide="ignore"):
        inv = np.where(d == 0, F(np.inf), F(1) / d).astype(F)
    best = np.finfo(F).max
    stack = [(root, F(-np.inf))]
    seq, deepest = [], 1
    while stack:
        link, t1 = stack.pop()
        if t1 > best:
            continue
        if link & 63:
            seq.append(link)
            t = tris_of(link, o, d)
            if t < best:
                best = t
            continue
        sl = _slots(nodes, link >> 6)
        boxes = np.array([b for b, _ in sl], F)
        with np.errstate(invalid="ignore"):
            a = (boxes[:, :3] - o) * inv
            c = (boxes[:, 3:] - o) * inv
        a = np.where(np.isnan(a), -np.inf, a)
        c = np.where(np.isnan(c), np.inf, c)
        lo, hi = np.minimum(a, c), np.maximum(a, c)
        e1 = np.maximum(np.maximum(lo[:, 0], 0), np.maximum(lo[:, 1], lo[:, 2]))
        e2 = np.minimum(np.minimum(hi[:, 0], best), np.minimum(hi[:, 1], hi[:, 2]))
        for k, (_, l) in enumerate(sl):
            if e1[k] <= e2[k]:
                stack.append((l, F(e1[k])))
        deepest = max(deepest, len(stack))
    return seq, best, deepest


@pytest.mark.parametrize("name", ["atrium:0.05", "atrium:0.1", "soup_5000", "two_clusters"])
def test_wide_tree_walk_reaches_the_same_leaves(name):
    host = _host(name)
    wide, wroot, wbound, absorbed = host.device_tree()
    lit, lroot, lbound, _ = host.device_tree(literal=True)
    info = host.info()
    bmin, bmax = np.array(list(info.bbox_min), F), np.array(list(info.bbox_max), F)
    # leaf triangles from the reference-layout arrays, decompressed against the leaf's box as found in the literal tree
    inner, pk = _leaf_tris(host)
    leaf_box = {}
    for n in range(lit.shape[0]):
        for box, l in _slots(lit, n):
            if l & 63:
                leaf_box[l] = box
    cache = {}

    def tris_of(link, o, d):
        if link not in cache:
            first, nreal = link >> 6, link & 63
            box = leaf_box[link]
            mn, size = box[:3], (box[3:] - box[:3]).astype(F)
            npk = (nreal + 7) // 8
            rel = pk[first:first + npk].astype(F) * (F(1) / F(65535))
            p = (np.float64(size)[None, None, :, None] * np.float64(rel) + np.float64(mn)[None, None, :, None]).astype(F)
            p = p.transpose(0, 3, 1, 2).reshape(npk * 8, 3, 3)[:nreal]
            cache[link] = (p[:, 0], (p[:, 1] - p[:, 0]).astype(F), (p[:, 2] - p[:, 0]).astype(F))
        v0, e1, e2 = cache[link]
        h = np.cross(d, e2)
        det = (e1 * h).sum(-1)
        with np.errstate(divide="ignore", invalid="ignore"):
            invd = 1.0 / det
            s = o - v0
            u = invd * (s * h).sum(-1)
            q = np.cross(s, e1)
            v = invd * (d * q).sum(-1)
            t = invd * (e2 * q).sum(-1)
            ok = (u >= 0) & (v >= 0) & (u + v <= 1) & (t >= 0)  # (inf - inf of a degenerate triangle compares false, as on the device)
        return F(np.where(ok, t, np.inf).min())

    rng = np.random.default_rng(5)
    ext = bmax - bmin
    n = 300
    o = (bmin - 0.2 * ext + rng.random((n, 3)) * ext * 1.4).astype(F)
    d = rng.standard_normal((n, 3)).astype(F)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # origins exactly on box planes of the tree, where entry distances are 0 or ties
    planes = lit[:, :, :6].view(F).reshape(-1, 6)
    planes = planes[np.isfinite(planes).all(1) & (lit[:, :, 6].reshape(-1) != NULL)]
    for k in range(0, 60):
        b = planes[rng.integers(0, planes.shape[0])]
        o[k] = b[:3] if k % 2 else b[3:]
    visited = 0
    for k in range(n):
        ws, wt, wd = _walk(wide, wroot, tris_of, o[k], d[k])
        ls, lt, ld = _walk(lit, lroot, tris_of, o[k], d[k])
        assert ws == ls, k
        assert np.array_equal(np.array([wt], F).view(np.uint32), np.array([lt], F).view(np.uint32))
        assert wd <= wbound and ld <= lbound
        visited += len(ws)
    assert visited > 50  # the rays do reach leaves
