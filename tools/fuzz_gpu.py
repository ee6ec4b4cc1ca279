#!/usr/bin/env python3
"""Differential fuzzing of the GPU render paths against the oracle: random small scenes, cameras (incl. axis-aligned views that
produce zero direction components), resolutions, tile sizes, sample counts, kernels (packets / groups / fused paths / staged paths),
work-unit sizes, progressive splits, material tables (grey, coloured, checker-textured) and sky radiance, instanced objects and object groups of different meshes, the
chunked accumulation rule.  Every
frame must match the oracle bit for bit.  usage: fuzz_gpu.py [cases] [seed] [cached]   ("cached": only cases that run the packet
walk's mask-cache kernel -- plain scenes, 16 to 256 samples, 4 to 32 samples in flight, lenses from f/0.05 to a pinhole, some cameras far away)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minipath_amd as mp
from oracle import pyoracle as po
from tests import meshes

def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)

CACHED = len(sys.argv) > 3 and sys.argv[3] == "cached"

def run(cases, seed, ctx=None):
    """Returns the number of mismatching cases."""
    rng = np.random.default_rng(seed)
    ctx = ctx if ctx is not None else mp.Context(0)
    scenes = {}
    for name in ("soup_300", "grid_40", "sphere_24", "flat_plane", "soup_5000"):
        pos, nrm, tex, tri = meshes.make(name)
        mat = (np.arange(tri.shape[0]) * 7 % 3).astype(np.uint32)  # three materials, interleaved
        scenes[name] = (mp.Scene(mp.TriangleBvh.build(pos, nrm, tex, tri, ctx, tri_material=mat)), po.Bvh.build(pos, nrm, tex, tri, tri_material=mat))
    from minipath_amd import scenes as _scenes   # ... and a small stand-in: thin top nodes, absorbed in the wide device tree
    pos, nrm, tex, tri = _scenes.atrium(1, 0.02)
    mat = (np.arange(tri.shape[0]) * 7 % 3).astype(np.uint32)
    scenes["atrium_0.02"] = (mp.Scene(mp.TriangleBvh.build(pos, nrm, tex, tri, ctx, tri_material=mat)), po.Bvh.build(pos, nrm, tex, tri, tri_material=mat))
    assert scenes["atrium_0.02"][0].object.device_tree()[3] > 0
    ball_def = ((0.3, 0.2, -0.1), 1.1)
    ball = mp.Sphere(*ball_def, ctx)
    bad = 0
    for case in range(cases):
        if case and case % 500 == 0:
            print(f"... {case} cases, {bad} mismatching so far", flush=True)
        name = list(scenes)[int(rng.integers(len(scenes)))]
        scene, ob = scenes[name]
        w, h = int(rng.integers(17, 150)), int(rng.integers(9, 120))
        ts = int(rng.choice([8, 16, 24, 32, 64]))
        spp = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 33, 70, 64, 130]))   # >= 64: units of four and more passes (mask cache)
        seed = int(rng.integers(1 << 40))
        if rng.random() < 0.3:   # axis-aligned view: exact zero direction components at the image centre lines
            axis = int(rng.integers(3)); eye = np.zeros(3); eye[axis] = float(rng.choice([-6.0, 6.0])); at = np.zeros(3)
            up = np.zeros(3); up[(axis + 1) % 3] = 1.0
        else:
            eye = rng.normal(size=3) * 4.0 + np.array([0, 1.0, 0]); at = rng.normal(size=3) * 0.5; up = np.array([0.0, 1.0, 0.0])
            if name.startswith("atrium"):   # inside the hall
                eye = np.array([rng.uniform(-15, 15), rng.uniform(1, 10), rng.uniform(-8, 8)]); at = eye + rng.normal(size=3)
        fnum = float(rng.choice([1.4, 4.8, 16.0, 1e9]))
        cam = mp.Camera.default().look_at(tuple(eye), tuple(at), tuple(up)).f_number(fnum)
        oc = po.Camera(); po.lib().mpo_camera_default(C.byref(oc)); po.lib().mpo_camera_look_at(C.byref(oc), po.vec3(*eye), po.vec3(*at), po.vec3(*up)); oc.f_number = fnum
        smp = po.build_sampler(oc, w, h)
        mode = str(rng.choice(["packets", "groups", "paths", "staged"]))
        if CACHED:   # only cases that run the packet walk's mask-cache kernel: plain scenes, units of four and more passes
            mode = "packets"; spp = int(rng.choice([16, 24, 33, 64, 70, 130, 256])); fnum = float(rng.choice([0.05, 0.7, 1.4, 4.8, 16.0, 1e9]))
            if rng.random() < 0.15:   # a far-away camera: long rays, small footprints
                eye = eye * float(rng.choice([30.0, 1000.0])); oc = po.Camera(); po.lib().mpo_camera_default(C.byref(oc)); po.lib().mpo_camera_look_at(C.byref(oc), po.vec3(*eye), po.vec3(*at), po.vec3(*up))
            cam = mp.Camera.default().look_at(tuple(eye), tuple(at), tuple(up)).f_number(fnum); oc.f_number = fnum
            smp = po.build_sampler(oc, w, h)
        depth = int(rng.integers(1, 6)) if mode in ("paths", "staged") else 0
        s_opt = int(rng.choice([0, 0, 1, 4, 8, 16, 32, 64]))
        ctx.set_option("packet_samples_in_flight", s_opt)
        ctx.set_option("packet_stack_registers", int(rng.choice([64, 64, 3, 9])))
        ctx.set_option("paths_pooled", int(rng.choice([0, 1, 2, 3, 3])))
        if CACHED:
            s_opt = int(rng.choice([0, 0, 4, 8, 16, 32])); ctx.set_option("packet_samples_in_flight", s_opt); ctx.set_option("packet_stack_registers", 64)
        ctx.set_option("packet_mask_cache", 1 if CACHED else int(rng.choice([0, 1, 2, 2])))   # the packet walk's per-unit child-rejection masks   # one pass per walk / pooled passes of the fused path kernel
        chunked = bool(rng.random() < 0.25)
        if chunked:
            spp = int(rng.choice([spp, 300, 513]))
        table = [(float(rng.uniform(0.1, 0.95)), float(rng.choice([0.0, 0.0, 2.5]))) for _ in range(3)]
        if not chunked and rng.random() < 0.4:   # coloured materials, checkerboards over HitRecord.texture_coords (never with the chunked rule)
            table = []
            for _ in range(3):
                e = {"albedo": [float(x) for x in rng.uniform(0.05, 0.95, 3)], "emission": [float(x) for x in rng.choice([0.0, 0.0, 1.5], 3)]}
                if rng.random() < 0.5:
                    e["albedo2"] = [float(x) for x in rng.uniform(0.05, 0.95, 3)]
                    e["checker"] = float(rng.choice([1.0, 3.0, 7.5, 40.0]))
                table.append(e)
        sky = float(rng.choice([1.0, 0.0, 0.4]))
        scene.object.set_materials(table, sky); ob.set_materials(table, sky)
        use = scene
        pick = 1.0 if CACHED else rng.random()
        if pick < 0.15:   # instanced object: 2-4 translated copies
            tr = (rng.normal(size=(int(rng.integers(2, 5)), 3)) * 3.0).astype(np.float32)
            use = mp.Scene(mp.Instances(scene.object, tr)); ob.set_instances(tr)
        elif pick < 0.3:   # object group: 2-4 members drawn from all the scenes (this one is the container)
            names = [name] + [list(scenes)[int(rng.integers(len(scenes)))] for _ in range(int(rng.integers(1, 4)))]
            order = rng.permutation(len(names))
            names = [names[i] for i in order]
            gm, om = [scenes[m][0].object for m in names], [scenes[m][1] for m in names]
            if rng.random() < 0.4:   # ... and a Sphere among them
                at_ = int(rng.integers(len(gm) + 1))
                gm.insert(at_, ball); om.insert(at_, ball_def)
            tr = (rng.normal(size=(len(gm), 3)) * 3.0).astype(np.float32)
            rot = None
            if rng.random() < 0.5:   # ... placed by rigid transforms (some of them exact identities / axis turns)
                rot = rng.standard_normal((len(gm), 4)).astype(np.float32)
                rot = (rot / np.linalg.norm(rot.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
                rot[0] = (0, 0, 0, 1)
                if len(gm) > 2: rot[2] = (0, np.float32(np.sqrt(0.5)), 0, np.float32(np.sqrt(0.5)))
            grp = mp.ObjectGroup(gm, tr, rotations=rot)
            grp.set_materials(table, sky)
            use = mp.Scene(grp); ob.set_group(om, tr, rotations=rot)
        else:
            ob.set_instances(np.zeros((0, 3), np.float32))
        po.lib().mpo_set_chunked_sum(1 if chunked else 0)
        st = mp.RenderSettings(ts, spp, (w, h), seed=seed, traversal="groups" if mode == "groups" else "packets", max_depth=depth, wavefront=(mode == "staged"),
                               chunked_sum=chunked)
        fr = mp.FrameRenderer(use, cam, st)
        split = bool(rng.random() < 0.4 and spp > 1)
        if split:   # progressive split
            cut = int(rng.integers(1, spp)); nxt = fr.render_pass(0, cut); gseg = int(fr.segments.item()); fr.render_pass(nxt); gseg += int(fr.segments.item())
        else:
            fr.render(); gseg = int(fr.segments.item())
        img, u8 = fr.untile()
        torch.cuda.synchronize()
        if depth:
            of, ou8, _, seg = ob.render_image_paths_mt(smp, w, h, spp, seed, depth, ts, 8)
        else:
            of, ou8, _, seg, _ = ob.render_image_mt(smp, w, h, spp, seed, ts, 8)
        po.lib().mpo_set_chunked_sum(0)
        ob.set_instances(np.zeros((0, 3), np.float32))
        ok = np.array_equal(bits(img.cpu().numpy()), bits(of)) and np.array_equal(u8.cpu().numpy(), ou8) and gseg == seg
        if not ok:
            bad += 1
            print(f"MISMATCH case {case}: {name} {w}x{h} ts{ts} spp{spp} seed{seed} mode {mode} depth {depth} S{s_opt} chunked {chunked} inst {use is not scene} sky {sky} eye{eye} at{at} f{fnum}: "
                  f"{int(np.sum(bits(img.cpu().numpy()) != bits(of)))} f32 values differ")
    ctx.set_option("packet_samples_in_flight", 0)
    ctx.set_option("packet_stack_registers", 64)
    ctx.set_option("paths_pooled", 1)
    ctx.set_option("packet_mask_cache", 1)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    bad = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n} cases, {bad} mismatching")
    sys.exit(1 if bad else 0)
