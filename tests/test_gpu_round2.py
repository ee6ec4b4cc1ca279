"""GPU parity tests (-m gpu) of the round-2 features, all through the C ABI, all bit-exact against the oracle:
mp_scene_from_arrays (import of reference-layout arrays), the material table of the build-defined path extension,
HitRecord.material, the chunked accumulation rule (MP_FLAG_CHUNKED_SUM), seed mixing."""
import numpy as np
import pytest

import minipath_amd as mp
from tests import meshes
from tests.conftest import TEAPOT

pytestmark = pytest.mark.gpu
SEED = 0x5EED


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return mp.Context(0)


@pytest.fixture(scope="module")
def teapot(ctx):
    return mp.Scene(mp.TriangleBvh.with_obj(TEAPOT, ctx))


def _render(scene, cam, st, tile=None):
    import torch

    fr = mp.FrameRenderer(scene, cam, st, tiles=None if tile is None else [mp.ScreenBlock(*tile)])
    fr.render()
    if tile is None:
        img, _ = fr.untile()
        torch.cuda.synchronize()
        return img.cpu().numpy(), int(fr.segments.item())
    torch.cuda.synchronize()
    tw, th = tile[2] - tile[0], tile[3] - tile[1]
    return fr.tile_buf[0, :th, :tw].cpu().numpy(), int(fr.segments.item())


def _permute_inside_leaves(inner, packets, shading, mat, seed):
    """What another sort_unstable_by_key (building.rs:295) could have produced: the real triangles of every leaf in another
    order (padding stays at the tail); topology, boxes and quantised vertices are untouched."""
    rng = np.random.default_rng(seed)
    pk = packets.copy().view(np.uint16).reshape(-1, 9, 8)       # [packet][vertex*3+coord][lane]
    sh, mt = shading.copy(), mat.copy()
    links = inner.view(np.uint32).reshape(-1, 32)[:, 24:].reshape(-1)
    moved = 0
    for link in links:
        cnt = int(link) & 7
        if int(link) == 0xFFFFFFF8 or cnt == 0:
            continue
        first = int(link) >> 3
        slots = np.arange(first * 8, (first + cnt) * 8)
        tri = pk[first:first + cnt].transpose(0, 2, 1).reshape(-1, 9)  # [slot][9]
        pad = np.all(tri == 0, axis=1) & np.all(sh[slots, :] == 0, axis=1)
        n_real = int(np.max(np.nonzero(~pad)[0])) + 1 if np.any(~pad) else 0
        perm = rng.permutation(n_real)
        moved += int(np.sum(perm != np.arange(n_real)))
        tri[:n_real] = tri[perm]
        sh[slots[:n_real]] = sh[slots[:n_real]][perm]
        mt[slots[:n_real]] = mt[slots[:n_real]][perm]
        pk[first:first + cnt] = tri.reshape(cnt, 8, 9).transpose(0, 2, 1)
    assert moved > 100
    return pk.view(np.uint8).reshape(-1, 144), sh, mt


def test_from_arrays_renders_bit_identical(ctx, teapot, oracle, teapot_oracle_bvh):
    """mp_scene_from_arrays: export -> import -> render equals the built scene bit for bit (frame, hits); a
    permuted-within-leaf variant -- the one documented deviation, building.rs:295 -- is rendered exactly as the oracle renders
    the SAME arrays, and differs from the original only in triangle_index values (and in exact-t ties, if any)."""
    import torch

    built = teapot.object
    i = built.info()
    inner, packets, shading, vn, vt, mat = built.export(with_material=True)
    again = mp.Scene(mp.TriangleBvh.from_arrays(inner, packets, shading, vn, vt, i.root_link, list(i.bbox_min), list(i.bbox_max), ctx, tri_material=mat))
    assert again.object.info().stack_bound == i.stack_bound
    cam = mp.Camera.teapot_view()
    st = mp.RenderSettings(64, 16, (256, 256), seed=SEED)
    a, _ = _render(teapot, cam, st)
    b, _ = _render(again, cam, st)
    assert np.array_equal(bits(a), bits(b))
    st8 = mp.RenderSettings(64, 8, (256, 256), seed=SEED, max_depth=6)
    a8, sa = _render(teapot, cam, st8)
    b8, sb = _render(again, cam, st8)
    assert np.array_equal(bits(a8), bits(b8)) and sa == sb

    # permuted inside the leaves
    p_packets, p_shading, p_mat = _permute_inside_leaves(inner, packets, shading, mat, 3)
    perm = mp.Scene(mp.TriangleBvh.from_arrays(inner, p_packets, p_shading, vn, vt, i.root_link, list(i.bbox_min), list(i.bbox_max), ctx, tri_material=p_mat))
    orc = oracle.Bvh.from_arrays(inner, p_packets, p_shading, vn, vt, i.root_link, list(i.bbox_min), list(i.bbox_max), material=p_mat)
    bmin, bmax = teapot_oracle_bvh.bbox()
    o, d = meshes.random_rays(20000, 9, bmin, bmax)
    got = perm.object.intersect(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda())
    torch.cuda.synchronize()
    t, prim, u, v = orc.trace(o, d)
    assert np.array_equal(got["prim"].cpu().numpy().view(np.uint32), prim)
    for k, e in (("t", t), ("u", u), ("v", v)):
        assert np.array_equal(bits(got[k].cpu().numpy()), bits(e)), k
    t0, prim0, _, _ = teapot_oracle_bvh.trace(o, d)
    assert np.array_equal(bits(t), bits(t0))                      # same closest distance whatever the lane order
    assert 0 < int(np.sum(prim != prim0))                         # ... under other triangle_index values
    c, _ = _render(perm, cam, st)
    osmp = oracle.build_sampler(oracle.teapot_camera(), 256, 256)
    of, _, _, _, _ = orc.render_image_mt(osmp, 256, 256, 16, SEED, 64, 8)
    assert np.array_equal(bits(c), bits(of))                      # GPU == oracle on the permuted arrays
    assert np.mean(bits(c) != bits(a)) < 1e-4                     # == the original frame except exact-t ties


def test_materials_and_hit_record_material(ctx, oracle):
    """material table {albedo, emission} indexed by TriangleShadingData.material, an emissive material as the
    light under a black sky; HitRecord.material (geometry/mod.rs:78) in mp_hits_soa.  GPU == oracle bit for bit, both the
    fused and the staged pipeline."""
    import torch

    pos, nrm, tex, tri = meshes.make("soup_5000")
    mat = (np.arange(tri.shape[0]) % 4).astype(np.uint32)
    scene = mp.Scene(mp.TriangleBvh.build(pos, nrm, tex, tri, ctx, tri_material=mat))
    orc = oracle.Bvh.build(pos, nrm, tex, tri, tri_material=mat)
    assert scene.object.info().material_count == 4
    assert np.array_equal(scene.object.export(with_material=True)[5], orc.tri_material())
    table = [(0.8, 0.0), (0.2, 0.0), (0.0, 6.0), (0.55, 0.25)]
    with pytest.raises(mp.MinipathError):
        scene.object.set_materials(table[:2])
    import ctypes as C

    oc = oracle.Camera()
    oracle.lib().mpo_camera_default(C.byref(oc))
    oracle.lib().mpo_camera_look_at(C.byref(oc), oracle.vec3(0.5, 0.2, 11.0), oracle.vec3(0, 0, 0), oracle.vec3(0, 1, 0))
    cam = mp.Camera.default().look_at((0.5, 0.2, 11.0), (0, 0, 0), (0, 1, 0))
    res = (128, 96)
    osmp = oracle.build_sampler(oc, *res)
    for sky in (0.0, 0.5):
        scene.object.set_materials(table, sky)
        orc.set_materials(table, sky)
        of, _, _, seg = orc.render_image_paths_mt(osmp, res[0], res[1], 7, 21, 5, 32, 8)
        for wavefront in (False, True):
            got, gseg = _render(scene, cam, mp.RenderSettings(32, 7, res, seed=21, max_depth=5, wavefront=wavefront))
            assert np.array_equal(bits(got), bits(of)), (sky, wavefront, int(np.sum(bits(got) != bits(of))))
            assert gseg == seg
        assert of[..., 0].max() > 1.0  # the light is visible
    # HitRecord.material
    bmin, bmax = orc.bbox()
    o, d = meshes.random_rays(5000, 2, bmin, bmax)
    got = scene.object.intersect(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), full=True)
    torch.cuda.synchronize()
    gm, gp = got["material"].cpu().numpy(), got["prim"].cpu().numpy().view(np.uint32)
    tm = orc.tri_material()
    hit = gp != 0xFFFFFFFF
    assert hit.sum() > 500 and np.array_equal(gm[hit].astype(np.uint32), tm[gp[hit]]) and np.all(gm[~hit] == 0)
    r = oracle.ray_new(o[hit][0], d[hit][0])
    assert orc.intersect(r).material == int(gm[hit][0])


@pytest.mark.parametrize("mode", ["packets", "groups", "paths", "staged"])
def test_chunked_sum_matches_oracle_and_any_pass_split(teapot, oracle, teapot_oracle_bvh, mode):
    """MP_FLAG_CHUNKED_SUM (build-defined accumulation rule for long sample chains, configs[4]): f32 sums over 256-sample chunks,
    f64 total.  GPU == oracle's statement of the same rule, bit for bit, in one launch and for ragged pass splits that cut
    chunks in the middle; and within 1e-5 of the reference's single-chain mean."""
    import torch

    cam = mp.Camera.teapot_view()
    res, spp, tile = (128, 128), 700, (48, 40, 80, 72)
    depth = 4 if mode in ("paths", "staged") else 0
    st = mp.RenderSettings(32, spp, res, seed=77, traversal="groups" if mode == "groups" else "packets", max_depth=depth,
                           wavefront=(mode == "staged"), chunked_sum=True)
    osmp = oracle.build_sampler(oracle.teapot_camera(), *res)
    oracle.lib().mpo_set_chunked_sum(1)
    try:
        if depth:
            of, _, _ = teapot_oracle_bvh.render_tile_paths(osmp, res[0], res[1], spp, 77, depth, *tile)
        else:
            of, _ = teapot_oracle_bvh.render_tile(osmp, res[0], res[1], spp, 77, *tile)
    finally:
        oracle.lib().mpo_set_chunked_sum(0)
    if depth:
        chain, _, _ = teapot_oracle_bvh.render_tile_paths(osmp, res[0], res[1], spp, 77, depth, *tile)
    else:
        chain, _ = teapot_oracle_bvh.render_tile(osmp, res[0], res[1], spp, 77, *tile)
    one, _ = _render(teapot, cam, st, tile)
    assert np.array_equal(bits(one), bits(of)), int(np.sum(bits(one) != bits(of)))
    assert np.max(np.abs(one - chain) / np.maximum(np.abs(chain), 1e-3)) <= 1e-5
    fr = mp.FrameRenderer(teapot, cam, st, tiles=[mp.ScreenBlock(*tile)])
    nxt = 0
    for count in (5, 246, 10, 300, 0):  # 5, 251, 261 (cuts chunk 1), 561 (cuts chunk 2), rest
        nxt = fr.render_pass(nxt, count)
    torch.cuda.synchronize()
    assert nxt == spp
    assert np.array_equal(bits(fr.tile_buf[0].cpu().numpy()), bits(of))


def test_consecutive_seeds_share_no_sample_rays(ctx):
    """with key = seed + index the frame of seed s+1 was the frame of seed s shifted by one sample.  The seed is
    now mixed (SplitMix64) before the index is added: the rays of seeds s and s+1 have nothing in common."""
    import ctypes as C

    import torch

    from minipath_amd import _lib

    cam = mp.Camera.teapot_view()
    blk = mp.ScreenBlock(10, 20, 42, 44)
    n = blk.area()
    rays = {}
    for seed in (100, 101):
        st = mp.RenderSettings(64, 16, (256, 256), seed=seed)
        s, ss = cam.build_sampler(st.resolution).as_struct(), st.as_struct()
        per_sample = []
        for sample in range(3):
            bufs = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(6)]
            _lib.check(_lib.lib().mp_generate_rays(ctx.handle, C.byref(s), C.byref(ss), blk.as_struct(), sample, *[b.data_ptr() for b in bufs], None))
            torch.cuda.synchronize()
            per_sample.append(np.stack([b.cpu().numpy() for b in bufs[3:]], -1))
        rays[seed] = np.stack(per_sample)  # [sample][pixel][dir xyz]
    a = {tuple(r) for r in bits(rays[100]).reshape(-1, 3).tolist()}
    b = {tuple(r) for r in bits(rays[101]).reshape(-1, 3).tolist()}
    assert len(a) == 3 * n and len(a & b) == 0


@pytest.mark.parametrize("n", [2, 3])
def test_multi_device_behind_the_c_abi(oracle, teapot_oracle_bvh, n):
    """N GPUs behind the boundary.  mp_render_begin_multi = render()'s worker pool with one host thread per device
    pulling from the one shared tile queue (machinery.rs:51-116, :205-208); mp_render_frame_multi = device-resident frame, rank
    r renders tiles r::n, shards gathered to device 0 by peer copies, un-tile there.  On this one-GPU box the n contexts share
    device 0 (the library allows it for exactly this purpose); the images must equal the oracle's frame bit for bit."""
    import threading

    import torch

    ctxs = [mp.Context(0) for _ in range(n)]
    scenes = [mp.Scene(mp.TriangleBvh.with_obj(TEAPOT, c)) for c in ctxs]
    cam = mp.Camera.teapot_view()
    res = (200, 136)
    st = mp.RenderSettings(16, 6, res, seed=SEED)
    of, ou8, *_ = teapot_oracle_bvh.render_image_mt(oracle.build_sampler(oracle.teapot_camera(), *res), res[0], res[1], 6, SEED, 16, 8)
    # render(): callbacks from n worker threads.  Every worker starts with a tile (machinery.rs:51-75): the first `started`
    # callback of each worker thread waits at a barrier for the other workers' first callbacks, so no worker can drain the
    # queue before the others have taken a batch (the queue holds at least 4 batches per worker); all n threads must show up
    lock, started, finished, threads = threading.Lock(), [], [], set()
    gate = threading.Barrier(n)

    def on_start(b):
        me = threading.get_ident()
        with lock:
            first = me not in threads
            threads.add(me)
            started.append(b)
        if first:
            gate.wait(timeout=60)

    def on_finish(b, snap):
        with lock:
            finished.append((b, snap.finished, snap.total))

    prog = mp.render_multi(scenes, cam, st, on_start, on_finish)
    prog.wait()
    assert prog.is_finished() and prog.progress().finished == prog.progress().total == 13 * 9
    assert np.array_equal(bits(prog.image_f32()), bits(of)) and np.array_equal(prog.image(), ou8)
    assert len(started) == len(finished) == 13 * 9 and len({(b.min_x, b.min_y) for b in started}) == 13 * 9
    assert sorted(f[1] for f in finished) == list(range(1, 13 * 9 + 1))
    assert len(threads) == n  # every device's worker took part (which worker gets which tile is a race, as in the reference)
    prog.close()
    # device-resident frame
    mf = mp.MultiDeviceFrame(scenes, cam, st)
    for _ in range(3):  # frames back to back reuse the shard / gather buffers
        img, img8 = mf.render()
    torch.cuda.synchronize()
    assert np.array_equal(bits(img.cpu().numpy()), bits(of)) and np.array_equal(img8.cpu().numpy(), ou8)
    assert mf.segments == res[0] * res[1] * 6
    # build-defined paths through the same entry point
    st4 = mp.RenderSettings(16, 5, res, seed=SEED, max_depth=4)
    pf, _, _, _ = teapot_oracle_bvh.render_image_paths_mt(oracle.build_sampler(oracle.teapot_camera(), *res), res[0], res[1], 5, SEED, 4, 16, 8)
    img4, _ = mp.MultiDeviceFrame(scenes, cam, st4).render()
    torch.cuda.synchronize()
    assert np.array_equal(bits(img4.cpu().numpy()), bits(pf))
    with pytest.raises(mp.MinipathError):
        mp.MultiDeviceFrame([scenes[0], scenes[0]], cam, st).render()  # the same context twice


def test_instanced_object(ctx, oracle, teapot_oracle_bvh):
    """SURVEY 8 f3 second half (multi-object / instancing behind `trait Object`, scene/mod.rs:7-10): translated instances of one
    TriangleBvh as the scene's object.  Build-defined (the reference has one object per Scene); the oracle states it operation by
    operation; GPU == oracle bit for bit for hits (incl. which instance), frames (both traversals request the group walk), paths."""
    import torch

    base = mp.TriangleBvh.with_obj(TEAPOT, ctx)
    tr = np.array([[0, 0, 0], [7.5, 0, -3], [-7.0, 0.5, -6], [0.25, 3.4, -1.0]], np.float32)
    inst = mp.Instances(base, tr)
    scene = mp.Scene(inst)
    orc = oracle.Bvh.from_obj(TEAPOT)
    orc.set_instances(tr)
    i = inst.info()
    assert i.triangle_count == 2256 and np.allclose(list(i.bbox_min), [-10, 0, -8]) and np.allclose(list(i.bbox_max)[:2], [10.92963, 6.55])
    # hits
    o, d = meshes.random_rays(30000, 4, np.array(list(i.bbox_min)), np.array(list(i.bbox_max)))
    got = inst.intersect(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), full=True)
    torch.cuda.synchronize()
    t, prim, u, v, which = orc.trace_inst(o, d)
    gp = got["prim"].cpu().numpy().view(np.uint32)
    assert np.array_equal(gp, prim) and np.array_equal(got["instance"].cpu().numpy().view(np.uint32), which)
    for k, e in (("t", t), ("u", u), ("v", v)):
        assert np.array_equal(bits(got[k].cpu().numpy()), bits(e)), k
    hit = prim != 0xFFFFFFFF
    assert sorted(np.unique(which[hit]).tolist()) == [0, 1, 2, 3]
    r = oracle.ray_new(o[hit][5], d[hit][5])
    h = orc.intersect(r)
    assert np.array_equal(bits(got["point"].cpu().numpy()[hit][5]), bits(np.array(list(h.point), np.float32)))  # world-space point
    assert np.array_equal(bits(got["normal"].cpu().numpy()[hit][5]), bits(np.array(list(h.normal), np.float32)))
    # frames: reference semantics and the path extension
    import ctypes as C

    oc = oracle.Camera()
    oracle.lib().mpo_camera_default(C.byref(oc))
    oracle.lib().mpo_camera_look_at(C.byref(oc), oracle.vec3(0, 6, 22), oracle.vec3(0, 2, -2), oracle.vec3(0, 1, 0))
    cam = mp.Camera.default().look_at((0, 6, 22), (0, 2, -2), (0, 1, 0))
    res = (192, 128)
    osmp = oracle.build_sampler(oc, *res)
    of, _, _, _, _ = orc.render_image_mt(osmp, res[0], res[1], 5, 3, 32, 8)
    for traversal in ("packets", "groups"):
        a, seg = _render(scene, cam, mp.RenderSettings(32, 5, res, seed=3, traversal=traversal))
        assert np.array_equal(bits(a), bits(of)), traversal
        assert seg == res[0] * res[1] * 5
    assert (of[..., 3] > 0).mean() > 0.15
    pf, _, _, pseg = orc.render_image_paths_mt(osmp, res[0], res[1], 4, 3, 5, 32, 8)
    b, gseg = _render(scene, cam, mp.RenderSettings(32, 4, res, seed=3, max_depth=5))
    assert np.array_equal(bits(b), bits(pf)) and gseg == pseg
    w, wseg = _render(scene, cam, mp.RenderSettings(32, 4, res, seed=3, max_depth=5, wavefront=True))  # round 3: the staged pipeline takes groups
    assert np.array_equal(bits(w), bits(pf)) and wseg == pseg
    with pytest.raises(mp.MinipathError):
        mp.Instances(inst, tr)  # instances of instances are not defined
    # one identity instance == the plain object
    one = mp.Scene(mp.Instances(base, [[0, 0, 0]]))
    x, _ = _render(one, mp.Camera.teapot_view(), mp.RenderSettings(32, 6, (128, 96), seed=9))
    y, _ = _render(mp.Scene(base), mp.Camera.teapot_view(), mp.RenderSettings(32, 6, (128, 96), seed=9))
    assert np.array_equal(bits(x), bits(y))


def test_object_group_of_different_meshes(ctx, oracle):
    """Multi-object scene ("a top-level list of {object, transform} behind the same Object entry points",
    scene/mod.rs:7-15): members are DIFFERENT objects (teapot, a triangle soup with three material ids -- twice --, a grid, and a
    Sphere, the reference's other Object), each with its own tree depth, vertex arrays and material ids; one material table for
    the group.  Build-defined;
    GPU == oracle bit for bit: full hit records with the member index and the member-local triangle index, frames at reference
    semantics, the path extension with emissive / dark materials and a black sky, chunked sums over split passes."""
    import ctypes as C

    import torch

    teapot = mp.TriangleBvh.with_obj(TEAPOT, ctx)
    o_teapot = oracle.Bvh.from_obj(TEAPOT)
    gpu, orc = [teapot], [o_teapot]
    for name in ("soup_300", "grid_40"):
        pos, nrm, tex, tri = meshes.make(name)
        mat = (np.arange(tri.shape[0]) * 5 % 3).astype(np.uint32) if name == "soup_300" else None
        gpu.append(mp.TriangleBvh.build(pos, nrm, tex, tri, ctx, tri_material=mat))
        orc.append(oracle.Bvh.build(pos, nrm, tex, tri, tri_material=mat))
    ball = mp.Sphere((0.0, 0.5, 0.0), 1.25, ctx)  # the reference's other Object (scene/primitives.rs:10-56) as a member
    gpu.append(ball)
    orc.append(((0.0, 0.5, 0.0), 1.25))
    members = [0, 1, 3, 2, 1]
    tr = np.array([[0, 0, 0], [5.5, 1.5, -1.0], [-2.5, 4.5, 2.0], [-6.0, 2.0, -2.5], [0.5, 5.5, -3.0]], np.float32)
    group = mp.ObjectGroup([gpu[k] for k in members], tr)
    scene = mp.Scene(group)
    table = [(0.8, 0.0), (0.2, 2.5), (0.6, 0.0)]
    group.set_materials(table, 0.25)
    box = oracle.Bvh.from_obj(TEAPOT)  # the container: its table and sky apply, its triangles are reached through member 0
    box.set_group([box if k == 0 else orc[k] for k in members], tr)
    box.set_materials(table, 0.25)
    i = group.info()
    assert i.triangle_count == sum(gpu[k].info().triangle_count for k in members) and i.material_count == 3
    assert i.depth == max(g.info().depth for g in gpu) and i.stack_bound == max(g.info().stack_bound for g in gpu)
    lo = np.min([np.array(list(gpu[k].info().bbox_min)) + tr[j] for j, k in enumerate(members)], axis=0)
    hi = np.max([np.array(list(gpu[k].info().bbox_max)) + tr[j] for j, k in enumerate(members)], axis=0)
    assert np.array_equal(np.array(list(i.bbox_min), np.float32), lo.astype(np.float32)) and np.array_equal(np.array(list(i.bbox_max), np.float32), hi.astype(np.float32))
    # hits: every field of the HitRecord, which member, member-local triangle
    o, d = meshes.random_rays(40000, 11, lo, hi)
    got = group.intersect(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), full=True)
    torch.cuda.synchronize()
    t, prim, u, v, which = box.trace_inst(o, d)
    assert np.array_equal(got["prim"].cpu().numpy().view(np.uint32), prim)
    assert np.array_equal(got["instance"].cpu().numpy().view(np.uint32), which)
    for k, e in (("t", t), ("u", u), ("v", v)):
        assert np.array_equal(bits(got[k].cpu().numpy()), bits(e)), k
    hit = prim != 0xFFFFFFFF
    assert sorted(np.unique(which[hit]).tolist()) == [0, 1, 2, 3, 4]
    on_ball = hit & (which == 2)
    assert on_ball.sum() > 300 and np.all(prim[on_ball] == 0) and np.all(got["material"].cpu().numpy()[on_ball] == 0)
    assert not got["tex"].cpu().numpy()[on_ball].any()
    idx = np.concatenate([np.flatnonzero(hit)[:: max(1, int(hit.sum()) // 400)], np.flatnonzero(on_ball)[:60]])
    mats = set()
    for j in idx:
        h = box.intersect(oracle.ray_new(o[j], d[j]))
        assert np.array_equal(bits(got["point"].cpu().numpy()[j]), bits(np.array(list(h.point), np.float32)))
        assert np.array_equal(bits(got["normal"].cpu().numpy()[j]), bits(np.array(list(h.normal), np.float32)))
        assert np.array_equal(bits(got["tex"].cpu().numpy()[j]), bits(np.array(list(h.tex), np.float32)))
        assert int(got["material"].cpu().numpy()[j]) == h.material
        mats.add(h.material)
    assert mats == {0, 1, 2}
    # frames
    eye, at = (1.0, 5.0, 19.0), (0.0, 2.5, -1.5)
    oc = oracle.Camera()
    oracle.lib().mpo_camera_default(C.byref(oc))
    oracle.lib().mpo_camera_look_at(C.byref(oc), oracle.vec3(*eye), oracle.vec3(*at), oracle.vec3(0, 1, 0))
    cam = mp.Camera.default().look_at(eye, at, (0, 1, 0))
    res = (176, 120)
    osmp = oracle.build_sampler(oc, *res)
    of, _, _, _, _ = box.render_image_mt(osmp, res[0], res[1], 6, 5, 32, 8)
    for traversal in ("packets", "groups"):
        a, seg = _render(scene, cam, mp.RenderSettings(32, 6, res, seed=5, traversal=traversal))
        assert np.array_equal(bits(a), bits(of)), traversal
        assert seg == res[0] * res[1] * 6
    assert (of[..., 3] > 0).mean() > 0.1
    for spp in (3, 8, 17):  # every samples-in-flight variant of the path kernel (1, 2, 4, 8)
        pf, _, _, pseg = box.render_image_paths_mt(osmp, res[0], res[1], spp, 5, 6, 32, 8)
        b, gseg = _render(scene, cam, mp.RenderSettings(32, spp, res, seed=5, max_depth=6))
        assert np.array_equal(bits(b), bits(pf)) and gseg == pseg, spp
        w, wseg = _render(scene, cam, mp.RenderSettings(32, spp, res, seed=5, max_depth=6, wavefront=True))  # round 3: staged pipeline on groups
        assert np.array_equal(bits(w), bits(pf)) and wseg == pseg, ("staged", spp)
    # 17 spp at 16 samples in flight and 3 spp one sample per lane: both instantiations of the group packet kernel (round 3)
    for spp in (17, 3):
        of2, _, _, _, _ = box.render_image_mt(osmp, res[0], res[1], spp, 5, 32, 8)
        a2, _ = _render(scene, cam, mp.RenderSettings(32, spp, res, seed=5))
        assert np.array_equal(bits(a2), bits(of2)), spp
    # chunked sums over split passes
    st = mp.RenderSettings(32, 600, (64, 40), seed=5, max_depth=3, chunked_sum=True)
    fr = mp.FrameRenderer(scene, mp.Camera.default().look_at(eye, at, (0, 1, 0)), st)
    nxt = 0
    for count in (100, 333, 0):
        nxt = fr.render_pass(nxt, count)
    img, _ = fr.untile()
    torch.cuda.synchronize()
    oc2 = oracle.build_sampler(oc, 64, 40)
    oracle.lib().mpo_set_chunked_sum(1)
    try:
        cf, _, _, _ = box.render_image_paths_mt(oc2, 64, 40, 600, 5, 3, 32, 8)
    finally:
        oracle.lib().mpo_set_chunked_sum(0)
    assert np.array_equal(bits(img.cpu().numpy()), bits(cf))
    # what is not defined
    with pytest.raises(mp.MinipathError):
        mp.ObjectGroup([group, teapot], [[0, 0, 0], [1, 0, 0]])       # groups do not nest
    with pytest.raises(mp.MinipathError):
        group.export()                                                # a group has no arrays of its own
    # a one-member group at the origin == the member itself
    one = mp.Scene(mp.ObjectGroup([gpu[1]], [[0, 0, 0]]))
    vcam = mp.Camera.default().look_at((0, 0, 9), (0, 0, 0), (0, 1, 0))
    x, _ = _render(one, vcam, mp.RenderSettings(32, 6, (128, 96), seed=9, max_depth=4))
    y, _ = _render(mp.Scene(gpu[1]), vcam, mp.RenderSettings(32, 6, (128, 96), seed=9, max_depth=4))
    assert np.array_equal(bits(x), bits(y))


def _unit_quaternions(rng, n):
    """Random rotations as nalgebra stores them (i, j, k, w), normalised in f32; the first one is the exact identity."""
    q = rng.standard_normal((n, 4)).astype(np.float32)
    q /= np.sqrt((q.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(np.float32)
    q[0] = (0.0, 0.0, 0.0, 1.0)
    return q.astype(np.float32)


def test_object_group_with_rotated_members(ctx, oracle):
    """Members placed by a rigid transform (rotation + translation, the Isometry3 of the reference's camera): world = q * local + t.
    The ray enters a member's frame through the inverse rotation without re-normalising its direction, the normal comes back
    through q; quaternion products are evaluated in the oracle's operation order on the device.  Full hit records (world-space
    normals and points), frames and paths == oracle, bit for bit; an identity quaternion member == the same member without
    rotations where the arithmetic is the same, and a 90-degree turn about y maps a hit's normal as expected."""
    import ctypes as C

    import torch

    rng = np.random.default_rng(17)
    teapot = mp.TriangleBvh.with_obj(TEAPOT, ctx)
    pos, nrm, tex, tri = meshes.make("soup_300")
    soup = mp.TriangleBvh.build(pos, nrm, tex, tri, ctx)
    ball = mp.Sphere((0.4, 0.0, 0.0), 1.0, ctx)
    o_teapot, o_soup = oracle.Bvh.from_obj(TEAPOT), oracle.Bvh.build(pos, nrm, tex, tri)
    gpu = [teapot, teapot, soup, ball, soup]
    tr = np.array([[0, 0, 0], [7.0, 0.5, -2.0], [-5.5, 2.0, 1.0], [0.0, 5.0, -1.0], [3.0, 4.5, 2.5]], np.float32)
    q = _unit_quaternions(rng, 5)
    group = mp.ObjectGroup(gpu, tr, rotations=q)
    scene = mp.Scene(group)
    box = oracle.Bvh.from_obj(TEAPOT)
    box.set_group([box, o_teapot, o_soup, ((0.4, 0.0, 0.0), 1.0), o_soup], tr, rotations=q)
    i = group.info()
    lo, hi = np.array(list(i.bbox_min), np.float32), np.array(list(i.bbox_max), np.float32)
    o, d = meshes.random_rays(40000, 23, lo, hi)
    got = group.intersect(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), full=True)
    torch.cuda.synchronize()
    t, prim, u, v, which = box.trace_inst(o, d)
    assert np.array_equal(got["prim"].cpu().numpy().view(np.uint32), prim)
    assert np.array_equal(got["instance"].cpu().numpy().view(np.uint32), which)
    for k, e in (("t", t), ("u", u), ("v", v)):
        assert np.array_equal(bits(got[k].cpu().numpy()), bits(e)), k
    hit = prim != 0xFFFFFFFF
    assert sorted(np.unique(which[hit]).tolist()) == [0, 1, 2, 3, 4] and hit.mean() > 0.1
    gn, gp_ = got["normal"].cpu().numpy(), got["point"].cpu().numpy()
    for j in np.flatnonzero(hit)[:: max(1, int(hit.sum()) // 500)]:
        h = box.intersect(oracle.ray_new(o[j], d[j]))
        assert np.array_equal(bits(gn[j]), bits(np.array(list(h.normal), np.float32))), (j, which[j])
        assert np.array_equal(bits(gp_[j]), bits(np.array(list(h.point), np.float32)))
    assert np.allclose(np.linalg.norm(gn[hit], axis=1), 1.0, atol=1e-5)  # rotations keep normals unit up to rounding
    # every hit lies inside the group's box (rotated corners + translation), up to rounding
    assert np.all(gp_[hit] >= lo - 1e-3) and np.all(gp_[hit] <= hi + 1e-3)
    # frames
    eye, at = (2.0, 6.0, 21.0), (0.5, 2.5, -0.5)
    oc = oracle.Camera()
    oracle.lib().mpo_camera_default(C.byref(oc))
    oracle.lib().mpo_camera_look_at(C.byref(oc), oracle.vec3(*eye), oracle.vec3(*at), oracle.vec3(0, 1, 0))
    cam = mp.Camera.default().look_at(eye, at, (0, 1, 0))
    res = (160, 112)
    osmp = oracle.build_sampler(oc, *res)
    of, _, _, _, _ = box.render_image_mt(osmp, res[0], res[1], 5, 8, 32, 8)
    a, seg = _render(scene, cam, mp.RenderSettings(32, 5, res, seed=8))
    assert np.array_equal(bits(a), bits(of)) and seg == res[0] * res[1] * 5 and (of[..., 3] > 0).mean() > 0.1
    pf, _, _, pseg = box.render_image_paths_mt(osmp, res[0], res[1], 9, 8, 5, 32, 8)
    b, gseg = _render(scene, cam, mp.RenderSettings(32, 9, res, seed=8, max_depth=5))
    assert np.array_equal(bits(b), bits(pf)) and gseg == pseg
    # a quarter turn about +y (q = (0, sin 45, 0, cos 45)): local +z faces world +x
    s45 = np.float32(np.sqrt(0.5))
    turned = mp.ObjectGroup([soup], [[0, 0, 0]], rotations=[[0, s45, 0, s45]])
    plain = mp.ObjectGroup([soup], [[0, 0, 0]])
    bi = soup.info()
    ro, rd = meshes.random_rays(4000, 31, np.array(list(bi.bbox_min)), np.array(list(bi.bbox_max)))   # rays in the local frame
    turn = lambda a: np.ascontiguousarray(np.stack([a[:, 2], a[:, 1], -a[:, 0]], axis=1))            # ... and the same rays turned
    hp = plain.intersect(torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda(), full=True)
    ht = turned.intersect(torch.from_numpy(turn(ro)).cuda(), torch.from_numpy(turn(rd)).cuda(), full=True)
    torch.cuda.synchronize()
    pp, pt = hp["prim"].cpu().numpy(), ht["prim"].cpu().numpy()
    same = (pp == pt) & (pp != -1)
    assert (pp != -1).sum() > 150 and same.sum() > 0.98 * (pp != -1).sum()  # sqrt(0.5) is rounded: a few grazing rays may differ
    assert np.allclose(hp["t"].cpu().numpy()[same], ht["t"].cpu().numpy()[same], rtol=1e-4, atol=1e-4)
    assert np.allclose(turn(hp["normal"].cpu().numpy()[same]), ht["normal"].cpu().numpy()[same], atol=1e-4)
    assert np.allclose(turn(hp["point"].cpu().numpy()[same]), ht["point"].cpu().numpy()[same], atol=1e-3)
    with pytest.raises(ValueError):
        mp.ObjectGroup([soup, soup], [[0, 0, 0], [1, 0, 0]], rotations=[[0, 0, 0, 1]])


def test_instances_keep_the_material_table_they_were_made_with(ctx, oracle):
    """Regression (found by tools/fuzz_gpu.py): an instanced scene shares its object's material table by reference; a later
    mp_scene_set_materials on the object gives the OBJECT a new table and must neither free the old one under the instanced scene
    nor be freed twice when the two are destroyed."""
    import gc

    pos, nrm, tex, tri = meshes.make("soup_300")
    mat = (np.arange(tri.shape[0]) % 2).astype(np.uint32)
    base = mp.TriangleBvh.build(pos, nrm, tex, tri, ctx, tri_material=mat)
    orc = oracle.Bvh.build(pos, nrm, tex, tri, tri_material=mat)
    cam = mp.Camera.default().look_at((0, 0, 9), (0, 0, 0), (0, 1, 0))
    import ctypes as C

    oc = oracle.Camera()
    oracle.lib().mpo_camera_default(C.byref(oc))
    oracle.lib().mpo_camera_look_at(C.byref(oc), oracle.vec3(0, 0, 9), oracle.vec3(0, 0, 0), oracle.vec3(0, 1, 0))
    osmp = oracle.build_sampler(oc, 64, 48)
    st = mp.RenderSettings(16, 4, (64, 48), seed=2, max_depth=3)
    t1, t2 = [(0.9, 0.0), (0.1, 3.0)], [(0.3, 1.0), (0.8, 0.0)]
    tr = np.array([[0, 0, 0], [2.5, 0.5, 0]], np.float32)
    base.set_materials(t1, 0.0)
    inst = mp.Instances(base, tr)           # made with table t1
    base.set_materials(t2, 1.0)             # the object moves on; the instanced scene must still show t1
    orc.set_instances(tr); orc.set_materials(t1, 0.0)
    of, _, _, _ = orc.render_image_paths_mt(osmp, 64, 48, 4, 2, 3, 16, 8)
    a, _ = _render(mp.Scene(inst), cam, st)
    assert np.array_equal(bits(a), bits(of))
    orc.set_instances(np.zeros((0, 3), np.float32)); orc.set_materials(t2, 1.0)
    of2, _, _, _ = orc.render_image_paths_mt(osmp, 64, 48, 4, 2, 3, 16, 8)
    b, _ = _render(mp.Scene(base), cam, st)
    assert np.array_equal(bits(b), bits(of2))
    inst.close(); base.set_materials(t1, 0.5); del inst; gc.collect()
    c, _ = _render(mp.Scene(base), cam, st)   # any double free above would surface here as a sticky HIP error
    assert c.shape == b.shape
