"""CPU: the oracle (and the product's host builder) still reproduce the committed golden vectors."""
import hashlib
import os

import numpy as np

import minipath_amd as mp
from tests.conftest import GOLDEN, TEAPOT


def test_oracle_reproduces_golden(oracle, teapot_oracle_bvh):
    g = np.load(os.path.join(GOLDEN, "teapot_golden.npz"))
    b = teapot_oracle_bvh
    assert list(g["bvh_counts"]) == [b.n_inner, b.n_packets, b.n_vertices, b.depth, b.root]
    assert g["bvh_sha256"][0] == hashlib.sha256(b.inner_nodes_bytes().tobytes()).hexdigest()
    assert g["bvh_sha256"][1] == hashlib.sha256(b.packets_bytes().tobytes()).hexdigest()
    assert g["bvh_sha256"][2] == hashlib.sha256(b.tri_shading().tobytes()).hexdigest()
    s = oracle.build_sampler(oracle.teapot_camera(), 256, 256)
    assert np.array_equal(s.as_array().view(np.uint32), g["sampler"].view(np.uint32))
    t, prim, u, v = b.trace(g["ray_o"], g["ray_d"])
    assert np.array_equal(prim, g["hit_prim"]) and np.array_equal(t.view(np.uint32), g["hit_t_bits"])
    f, u8 = b.render_tile(s, 256, 256, 16, int(g["seed"]), *[int(x) for x in g["tile"]])
    assert np.array_equal(f.view(np.uint32), g["tile_f32_bits"]) and np.array_equal(u8, g["tile_u8"])


def test_product_builder_reproduces_golden_digest():
    g = np.load(os.path.join(GOLDEN, "teapot_golden.npz"))
    inner, packets, shading, _, _ = mp.TriangleBvh.with_obj(TEAPOT).export()
    assert g["bvh_sha256"][0] == hashlib.sha256(inner.tobytes()).hexdigest()
    assert g["bvh_sha256"][1] == hashlib.sha256(packets.tobytes()).hexdigest()
    assert g["bvh_sha256"][2] == hashlib.sha256(shading.tobytes()).hexdigest()


def test_oracle_traversal_matches_brute_force(oracle, teapot_oracle_bvh):
    """Self-check of the restated traversal (ray_bvh_intersection.rs:26-140), independent of BVH topology: the
    closest hit over ALL decompressed triangles tested one by one (numpy f64 Moller-Trumbore on the oracle's own
    quantised vertices) has the same t (to f32 rounding) and hits/misses agree away from silhouettes."""
    b = teapot_oracle_bvh
    g = np.load(os.path.join(GOLDEN, "teapot_golden.npz"))
    o, d = g["ray_o"][:400].astype(np.float64), g["ray_d"][:400].astype(np.float64)
    # decompress every packet against its leaf box by walking the tree with the oracle's box chain
    inner = b.inner_nodes_bytes().view(np.uint8).reshape(-1, 128)
    packets = b.packets_bytes().view(np.uint16).reshape(-1, 3, 3, 8)
    inv = np.float32(1.0) / np.float32(65535.0)
    tris = []

    def dec(q, size, mn):
        return (np.float32(size) * (q.astype(np.float32) * inv) + np.float32(mn)).astype(np.float32)

    def walk(link, mn, mx):
        size = (mx - mn).astype(np.float32)
        if link == 0xFFFFFFF8:
            return
        idx, cnt = link >> 3, link & 7
        if cnt == 0:
            node = inner[idx]
            u16 = node[:96].view(np.uint16).reshape(2, 3, 8)
            links = node[96:].view(np.uint32)
            for i in range(8):
                cmn = np.array([dec(u16[0, k, i], size[k], mn[k]) for k in range(3)], np.float32)
                cmx = np.array([dec(u16[1, k, i], size[k], mn[k]) for k in range(3)], np.float32)
                walk(int(links[i]), cmn, cmx)
        else:
            for p in range(idx, idx + cnt):
                for lane in range(8):
                    v = np.array([[dec(packets[p, a, k, lane], size[k], mn[k]) for k in range(3)] for a in range(3)], np.float64)
                    if not np.all(packets[p, :, :, lane] == 0):
                        tris.append(v)

    bmin, bmax = b.bbox()
    walk(b.root, bmin, bmax)
    T = np.array(tris)
    assert len(T) >= 2256
    v0, e1, e2 = T[:, 0], T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]
    t_oracle, prim, _, _ = b.trace(g["ray_o"][:400], g["ray_d"][:400])
    agree = 0
    for i in range(400):
        dn = d[i] / np.linalg.norm(d[i])
        h = np.cross(dn, e2)
        det = np.einsum("ij,ij->i", e1, h)
        with np.errstate(divide="ignore", invalid="ignore"):
            invd = 1.0 / det
            s = o[i] - v0
            u = invd * np.einsum("ij,ij->i", s, h)
            q = np.cross(s, e1)
            v = invd * (q @ dn)
            t = invd * np.einsum("ij,ij->i", e2, q)
        ok = (u >= 0) & (v >= 0) & (u + v <= 1) & (t >= 0)
        tb = t[ok].min() if ok.any() else None
        if tb is None:
            agree += prim[i] == 0xFFFFFFFF
        elif prim[i] != 0xFFFFFFFF:
            assert abs(tb - t_oracle[i]) <= 2e-5 * max(1.0, tb), (i, tb, t_oracle[i])
            agree += 1
    assert agree >= 396  # silhouette rays may flip between f32 and f64


def test_chunked_sum_rule_is_close_to_the_single_chain(oracle, teapot_oracle_bvh):
    """BUILD-DEFINED accumulation rule for long sample chains (configs[4]; include/minipath_hip.h MP_FLAG_CHUNKED_SUM):
    f32 sums over 256-sample chunks + f64 total.  It must agree with the reference's single f32 chain (worker.rs:40-44) to
    <= 1e-5 relative at sample counts where that chain is still accurate, and both must agree with an f64 sum."""
    b = teapot_oracle_bvh
    s = oracle.build_sampler(oracle.teapot_camera(), 256, 256)
    tile = (120, 100, 128, 108)
    spp = 1500  # five full chunks and a ragged one
    chain, _ = b.render_tile(s, 256, 256, spp, 7, *tile)
    oracle.lib().mpo_set_chunked_sum(1)
    try:
        chunked, _ = b.render_tile(s, 256, 256, spp, 7, *tile)
    finally:
        oracle.lib().mpo_set_chunked_sum(0)
    # f64 reference sum of the same per-sample values
    import ctypes as C
    exact = np.zeros((8, 8, 4))
    rgba = (C.c_float * 4)()
    for y in range(8):
        for x in range(8):
            for i in range(spp):
                oracle.lib().mpo_render_sample(b.h, C.byref(s), 256, spp, C.c_uint64(7), tile[0] + x, tile[1] + y, i, rgba, None)
                exact[y, x] += np.array(list(rgba), np.float64)
    exact /= spp
    scale = np.maximum(np.abs(exact), 1e-3)
    assert np.max(np.abs(chunked - exact) / scale) <= 1e-6  # 256-term f32 chunks + one f32 rounding of the mean; independent of spp
    assert np.max(np.abs(chain - exact) / scale) <= 1e-5
    assert np.max(np.abs(chain - chunked) / scale) <= 1e-5
    assert chunked[..., 3].max() == 1.0 and np.array_equal(chunked[..., 3], chain[..., 3])  # hit counts are exact either way


def test_materials_defaults_and_emissive_light(oracle):
    """BUILD-DEFINED path extension with a material table: the defaults {0.75, 0} + sky 1 are the round-1
    definition; with sky 0 only paths that reach an emissive triangle carry radiance."""
    from tests import meshes

    pos, nrm, tex, tri = meshes.make("soup_300")
    mat = (np.arange(tri.shape[0]) % 3).astype(np.uint32)
    plain = oracle.Bvh.build(pos, nrm, tex, tri)
    withm = oracle.Bvh.build(pos, nrm, tex, tri, tri_material=mat)
    assert withm.material_count == 3 and plain.material_count == 1
    assert np.array_equal(plain.packets_bytes(), withm.packets_bytes())  # ids ride along, geometry unchanged
    cam = oracle.Camera()
    oracle.lib().mpo_camera_default(cam)
    oracle.lib().mpo_camera_look_at(cam, oracle.vec3(0, 0, 9), oracle.vec3(0, 0, 0), oracle.vec3(0, 1, 0))
    s = oracle.build_sampler(cam, 64, 64)
    withm.set_materials([(0.75, 0.0)] * 3, 1.0)
    a, _, sa = plain.render_tile_paths(s, 64, 64, 4, 3, 5, 16, 16, 48, 48)
    b, _, sb = withm.render_tile_paths(s, 64, 64, 4, 3, 5, 16, 16, 48, 48)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa == sb
    withm.set_materials([(0.5, 0.0), (0.0, 4.0), (0.9, 0.0)], 0.0)
    c, _, sc = withm.render_tile_paths(s, 64, 64, 4, 3, 5, 16, 16, 48, 48)
    assert sc == sa                       # same geometry, same RNG stream: same segments
    assert c.max() > 0.5 and c.min() == 0.0 and not np.array_equal(a, c)
    assert np.all(c[a[..., 3] == 0][..., 0] == 0.0)  # primary miss under a black sky carries nothing
    try:
        withm.set_materials([(0.5, 0.0)], 1.0)
        raise AssertionError("table shorter than the ids in use must be refused")
    except RuntimeError:
        pass


def test_coloured_and_checker_materials_definition(oracle):
    """SURVEY 8 f4 finished: rgb albedo / emission and a material that READS HitRecord.texture_coords
    (geometry/mod.rs:78-79) -- a procedural checkerboard.  Build-defined, parity unpinned (the reference never reads either field);
    checked here against the definition itself: a grey table gives r = g = b and the old bits; per-channel tables render each
    channel exactly like a grey table holding that channel's values; a checker material with albedo2 == albedo is the untextured
    material; with different colours the two cells alternate along the grid's texture coordinates."""
    from tests import meshes

    pos, nrm, tex, tri = meshes.make("grid_40")            # uv in [0,1]^2 over the height field
    mat = (np.arange(tri.shape[0]) % 2).astype(np.uint32)
    b = oracle.Bvh.build(pos, nrm, tex, tri, tri_material=mat)
    cam = oracle.Camera()
    oracle.lib().mpo_camera_default(cam)
    oracle.lib().mpo_camera_look_at(cam, oracle.vec3(0.3, 6.0, 4.0), oracle.vec3(0, 0, 0), oracle.vec3(0, 1, 0))
    s = oracle.build_sampler(cam, 96, 64)
    args = (s, 96, 64, 5, 77, 4, 16, 8, 80, 56)
    b.set_materials([(0.6, 0.0), (0.3, 0.5)], 0.7)
    grey, _, seg = b.render_tile_paths(*args)
    assert np.array_equal(grey[..., 0], grey[..., 1]) and np.array_equal(grey[..., 0], grey[..., 2])
    rgb_table = [((0.6, 0.2, 0.9), (0.0, 0.1, 0.0)), ((0.3, 0.8, 0.1), (0.5, 0.0, 2.0))]
    b.set_materials(rgb_table, 0.7)
    col, _, seg2 = b.render_tile_paths(*args)
    assert seg2 == seg and np.array_equal(col[..., 3], grey[..., 3])
    for c in range(3):   # channel c of the coloured render == a grey render with that channel's numbers
        b.set_materials([(a[c], e[c]) for a, e in rgb_table], 0.7)
        one, _, _ = b.render_tile_paths(*args)
        assert np.array_equal(col[..., c].view(np.uint32), one[..., 0].view(np.uint32)), c
    assert not np.array_equal(col[..., 0], col[..., 1])
    # checker with equal colours == no texture
    b.set_materials([{"albedo": 0.6, "albedo2": 0.6, "checker": 9.0}, (0.3, 0.5)], 0.7)
    same, _, _ = b.render_tile_paths(*args)
    assert np.array_equal(same.view(np.uint32), grey.view(np.uint32))
    # a real checkerboard on material 0 under a bright sky, depth 1 view of it: both colours show up
    b.set_materials([{"albedo": (0.9, 0.9, 0.9), "albedo2": (0.1, 0.1, 0.8), "checker": 8.0}, {"albedo": 0.5, "albedo2": (0.9, 0.1, 0.1), "checker": 8.0}], 1.0)
    chk, _, _ = b.render_tile_paths(s, 96, 64, 3, 77, 2, 16, 8, 80, 56)
    assert not np.array_equal(chk[..., 0], chk[..., 2])
    hit = chk[..., 3] > 0
    assert hit.any() and (chk[hit][:, 2] > chk[hit][:, 0] + 0.2).any() and (chk[hit][:, 0] > chk[hit][:, 2] + 0.2).any()
