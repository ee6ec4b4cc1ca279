#!/usr/bin/env python3
"""bench.py -- Mrays/s of the per-pixel sampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one pass of the hot path over one frame of synthetic input: every pixel x every sample of the workload is
generated, traced, shaded and accumulated on the GPU (one launch per rank), the per-rank tile shards are gathered
to rank 0 over RCCL (N > 1) and scattered into the image-major framebuffer.  Scene (BVH) and camera are resident in
HBM before the timed region; the framebuffer stays in HBM.

Workload at N = 1 = the configuration BASELINE.json's metric is quoted on ("Sponza 1080p 256spp"): 1920x1080, 256 spp,
tile 64, seed 0x5EED.  `data/Sponza` is an empty, un-fetched submodule of the reference (.gitmodules:1-3), so the scene is
the seeded procedural stand-in `minipath_amd.scenes.atrium(seed=1)` (258 432 triangles, 28-level BVH8), built by the
reference's own builder algorithm; `config.workload` says so.  `value` is measured with the REFERENCE SEMANTICS (primary
ray + |d.n| shading = depth 1: the reference has no bounce loop): a "ray" is one Object::intersect call, so
rays == samples.  Secondary measurements in the same run (outside the timed region, own keys in the JSON line):
  "teapot_c2"    BASELINE configs[1]: teapot.obj 1920x1080 256 spp, reference semantics and the build-defined depth-8 paths
  "paths_depth8" BASELINE configs[2]: the stand-in at 1920x1080 64 spp with the build-defined max-depth-8 path extension
                 (rays = traced path segments)

Multi-GPU (--gpus N): one process per GPU.  Launched by torch.distributed.run (WORLD_SIZE set) the ranks run as they are;
started as plain `python bench.py --gpus N`, the parent starts the N ranks itself (child processes, before it touches the
GPU) and relays rank 0's JSON line and exit code.  Tiles are sharded round-robin over the ranks, the total work is fixed
("strong" scaling), no collective on the data path except the final framebuffer gather (pipelined one frame deep).

After the warmup frames every rank orders the hand-out of ITS tiles to its waves by the cost the warmup measured (expensive
tiles first; mp_launch_extras / FrameRenderer.rebalance).  The work and the image are unchanged (bit-identical, tested).

roofline: SURVEY 8(d) prices the path as a wavefront pipeline with 136 algorithmic HBM bytes per depth-1 ray (320 per bounce
segment); `achieved` / `frac` follow that definition and are MODELLED bytes over measured kernel time ("modelled": true).
The fused kernels keep ray state in registers, so their real HBM traffic is far smaller: `traffic` (bytes per launch) and
`hbm_measured_gbs` come from rocprofv3 PMC counters of the same workload (profiles/r03_counters.json; FETCH_SIZE x 2 +
WRITE_SIZE, the guide's gfx950 correction), and the actual limiter is reported as `valu_issue_frac` = SQ_INSTS_VALU x 2 cycles
/ (SIMDs x 2.4 GHz x kernel time) and `salu_issue_frac` = SQ_INSTS_SALU / (CUs x 2.4 GHz x kernel time), both with the live
kernel time of this run.  The counters file records the SHA-256 of the device sources it was collected from; when the tree's
differ, the line carries "counters_stale": true and none of the counter-derived figures.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_RAY_DEPTH1 = 136  # algorithmic HBM bytes per ray segment of the depth-1 wavefront formulation
B_RAY_BOUNCE = 320  # compacted wavefront bounce mode (build-defined path extension, --depth N)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
SHADER_CLOCK_HZ = 2.4e9  # MI355X_MICROARCH.md: peak engine clock
VALU_CYCLES = 2  # wave64 VALU instruction on a SIMD-32: 2 issue cycles (MI355X_MICROARCH.md "Per-instruction cycle constants")
TEAPOT = os.path.join(ROOT, "tests", "golden", "teapot.obj")
COUNTERS = os.path.join(ROOT, "profiles", "r03_counters.json")
# sources that decide what the device executes; their SHA-256 is stored with the counters (tools/collect_counters.py)
DEVICE_SOURCES = ["minipath_amd/csrc/kernels.hip", "minipath_amd/csrc/mp_internal.h", "minipath_amd/csrc/device_tree.cpp"]


def device_sources_sha256():
    import hashlib

    h = hashlib.sha256()
    for rel in DEVICE_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--scene", default="atrium", help="'atrium' (Sponza stand-in, default), 'teapot', or a path to an .obj")
    ap.add_argument("--detail", type=float, default=1.0, help="atrium tessellation (1.0 = 258 k triangles)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tile-stride", type=int, default=11, help="cpu_baseline renders every k-th tile")
    ap.add_argument("--cpu-threads", type=int, default=16, help="upper bound on cpu_baseline worker threads")
    ap.add_argument("--traversal", default="packets", choices=["packets", "groups"])
    ap.add_argument("--depth", type=int, default=0, help="0 = reference semantics (default); N >= 1 = build-defined path extension with at most N segments")
    ap.add_argument("--balance", default="static", choices=["static", "lpt"],
                    help="N > 1: 'lpt' re-partitions the tiles over the ranks by the cost the warmup frames measured (opt-in)")
    ap.add_argument("--wavefront", action="store_true", help="with --depth N: staged evaluation (HBM path streams, sorted)")
    ap.add_argument("--progressive", type=int, default=1, metavar="P",
                    help="render every frame as P progressive passes of spp/P samples (BASELINE configs[4]'s shape): the running sums "
                         "stay sharded on their GPUs, the framebuffer is gathered once per frame, after the last pass")
    ap.add_argument("--no-extension", action="store_true", help="skip the secondary measurements (teapot_c2, paths_depth8)")
    ap.add_argument("--check", action="store_true", help="compare a few tiles of the GPU frame with the oracle")
    args = ap.parse_args(argv)
    if args.scene == "teapot":
        args.scene = TEAPOT
    return args


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (one per GPU) and relay rank 0's line.
    Runs before anything in this process touches the GPU; the ranks are child processes, never an exec."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def workload_key(scene_name, w, h, spp, tile, depth, traversal):
    return f"{scene_name} {w}x{h} {spp}spp tile{tile} depth{depth} {traversal}"


def load_counters(key):
    """(counters of the workload or None, stale): stale = the counters were collected from other device sources than the tree's
    (SHA-256 over DEVICE_SOURCES differs): they are then NOT used for valu_issue_frac / salu_issue_frac / traffic."""
    try:
        d = json.load(open(COUNTERS))
    except Exception:
        return None, False
    c = d.get(key)
    if c is None:
        return None, False
    return c, d.get("_meta", {}).get("device_sources_sha256") != device_sources_sha256()


def make_scene(mp, ctx, scene_arg, detail):
    """Returns (scene, camera, short name, description)."""
    if scene_arg == "atrium":
        from minipath_amd import scenes

        bvh = mp.TriangleBvh.build(*scenes.atrium(1, detail), ctx)
        n = bvh.info().triangle_count
        return (mp.Scene(bvh), scenes.atrium_camera(), "atrium",
                f"atrium stand-in for the absent data/Sponza (reference .gitmodules:1-3: un-fetched submodule), {n} triangles, "
                f"generator seed 1" + ("" if detail == 1.0 else f", detail {detail}"))
    bvh = mp.TriangleBvh.with_obj(scene_arg, ctx)
    return mp.Scene(bvh), mp.Camera.teapot_view(), os.path.basename(scene_arg), f"{os.path.basename(scene_arg)} ({bvh.info().triangle_count} triangles), view of benches/render_teapot.rs:12-19"


def oracle_scene(po, mp, scene_arg, detail):
    """The CPU oracle's copy of the scene.  For the stand-in the oracle takes the reference-layout arrays of the product's
    builder (byte-identical to the oracle's own, slower, restated builder: tests/test_host_cpu.py) -- only the traversal is timed."""
    if scene_arg == "atrium":
        import ctypes as C

        from minipath_amd import scenes

        host = mp.TriangleBvh.build(*scenes.atrium(1, detail))  # host-only build (no GPU)
        i = host.info()
        inner, packets, shading, vn, vt, mat = host.export(with_material=True)
        b = po.Bvh.from_arrays(inner, packets, shading, vn, vt, i.root_link, list(i.bbox_min), list(i.bbox_max), material=mat)
        cam = po.Camera()
        po.lib().mpo_camera_default(C.byref(cam))
        eye, at, fnum = scenes.ATRIUM_VIEW
        po.lib().mpo_camera_look_at(C.byref(cam), po.vec3(*eye), po.vec3(*at), po.vec3(0, 1, 0))
        cam.f_number = fnum
        return b, cam
    return po.Bvh.from_obj(scene_arg), po.teapot_camera()


def cpu_baseline(args, mp):
    """The oracle (C restatement of the reference CPU path: one thread per core, atomic tile queue,
    machinery.rs:31-116) timed on this box's host cores on a bounded sample of the SAME workload."""
    from oracle import pyoracle as po

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, args.cpu_threads)  # `cores` = the threads actually used
    b, cam = oracle_scene(po, mp, args.scene, args.detail)
    s = po.build_sampler(cam, args.width, args.height)
    ntiles = len(po.tile_ordering(0, 0, args.width, args.height, args.tile))
    stride = max(1, args.cpu_tile_stride)
    if args.depth > 0:
        _, _, secs, rays = b.render_image_paths_mt(s, args.width, args.height, args.spp, args.seed, args.depth, args.tile, cores, 0, stride)
    else:
        _, _, secs, rays, _ = b.render_image_mt(s, args.width, args.height, args.spp, args.seed, args.tile, cores, 0, stride)
    return {
        "value": rays / secs / 1e6,
        "unit": "Mrays/s",
        "cores": cores,
        "kind": "port",
        "sample": f"every {stride}th of the {ntiles} {args.tile}x{args.tile} tiles of the same frame at full {args.spp} spp "
                  f"({rays / 1e6:.1f} Mrays, {secs:.1f} s wall on {cores} threads); C restatement of the reference CPU path "
                  "(the 8-lane decompression, slab and triangle loops are written so that gcc -O3 -mavx2 -mfma turns each into 256-bit AVX2 "
                  "code like the reference's f32x8; one thread per core on an atomic tile queue), "
                  "not the Rust reference (no Rust toolchain in this pipeline)",
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    import minipath_amd as mp

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # MP_BENCH_REHEARSAL=1: several ranks share one GPU over gloo (RCCL refuses duplicate devices) -- exercises the N > 1
    # code path on a one-GPU box; the numbers of such a run mean nothing and the JSON line says so
    rehearsal = os.environ.get("MP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from minipath_amd.distributed import DistributedFrame

    ctx = mp.Context(local_rank)
    scene, cam, scene_name, scene_desc = make_scene(mp, ctx, args.scene, args.detail)
    st = mp.RenderSettings(args.tile, args.spp, (args.width, args.height), seed=args.seed, traversal=args.traversal, max_depth=args.depth,
                           wavefront=args.wavefront)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def measure(scene_, cam_, st_, steps, warmup, balance="static"):
        """W untimed + exactly K timed frames bracketed by barrier + synchronize.  Returns a dict of whole-job numbers."""
        frame = DistributedFrame(scene_, cam_, st_, rank, world)
        events = []
        seg_acc = None
        for _ in range(warmup):
            frame.step(want_u8=True)
        frame.flush()
        if warmup > 0 and balance == "lpt" and world > 1:
            frame.repartition_by_cost()  # cost-balanced shards, expensive tiles first (collective)
        elif warmup > 0:
            frame.rebalance()  # hand the tiles the warmup frames found expensive to the waves first (same image, shorter tail)
        barrier()
        t0 = time.perf_counter()
        img = None
        for _ in range(steps):
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            if args.progressive > 1:  # P passes into the resident shard, one gather + un-tile at the end of the frame
                spp_, per = st_.sample_count, -(-st_.sample_count // args.progressive)
                ev[0].record()
                nxt = 0
                seg_acc = torch.zeros(1, dtype=torch.int64, device=dev)
                while nxt < spp_:
                    nxt = frame.render_pass(nxt, min(per, spp_ - nxt) if nxt + per < spp_ else 0)
                    seg_acc += frame.renderer.segments  # on the stream: no host synchronisation between the passes
                ev[1].record()
                out_img, _ = frame.gather_image(spp_, want_u8=True)
            else:
                out_img, _ = frame.step(want_u8=True, kernel_events=ev)  # N > 1: the previous frame's image (pipelined gather)
            events.append(ev)
            img = out_img if out_img is not None else img
        last, _ = frame.flush()  # completes the last frame's gather + un-tile inside the timed region
        img = last if last is not None else img
        barrier()
        elapsed = time.perf_counter() - t0
        k_ms = sum(a.elapsed_time(b) for a, b in events) / max(len(events), 1)  # HIP events on the launch stream
        seg_local = int((seg_acc if args.progressive > 1 else frame.renderer.segments).item())  # Object::intersect calls of this rank's last frame
        if world > 1:
            red = torch.tensor([elapsed, k_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
            elapsed, k_ms = float(red[0].item()), float(red[1].item())
            segt = torch.tensor([seg_local], dtype=torch.int64, device=dev)
            dist.all_reduce(segt, op=dist.ReduceOp.SUM)
            seg_total = int(segt.item())
        else:
            seg_total = seg_local
        return {"elapsed": elapsed, "kernel_ms": k_ms, "seg_local": seg_local, "seg_total": seg_total, "img": img, "tiles": frame.all_tiles}

    def roofline(kernel, m, b_ray, key, cu_count):
        """SURVEY 8(d) algorithmic figure + the measured counters of the same workload (profiles/r03_counters.json), the latter
        only if they were collected from the device sources of this tree (otherwise "counters_stale": true and no derived figures)."""
        k_s = m["kernel_ms"] * 1e-3
        achieved = m["seg_local"] * b_ray / k_s / 1e9
        r = {
            "bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "modelled": True,
            "achieved_is": f"{b_ray} algorithmic B/ray of SURVEY 8(d)'s wavefront formulation x rays per launch / measured kernel time -- "
                           "NOT counter bandwidth (the fused kernel keeps ray state in registers)",
            "kernel_ms": m["kernel_ms"], "algorithmic_bytes_per_launch": m["seg_local"] * b_ray, "bytes_per_ray": b_ray,
            "traffic": None,
        }
        c, stale = load_counters(key) if world == 1 else (None, False)
        if c and stale:
            r["counters_stale"] = True  # profiles/r03_counters.json predates the device sources: nothing derived from it
            c = None
        if c:
            r["traffic"] = c.get("hbm_bytes_per_launch")
            if r["traffic"]:
                r["hbm_measured_gbs"] = r["traffic"] / k_s / 1e9
                r["hbm_measured_frac"] = r["hbm_measured_gbs"] / HBM_PEAK_GBS
            if c.get("SQ_INSTS_VALU"):
                simds = cu_count * 4
                r["limiter"] = "valu_issue"
                r["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]
                r["valu_issue_frac"] = c["SQ_INSTS_VALU"] * VALU_CYCLES / (simds * SHADER_CLOCK_HZ * k_s)
                if c.get("valu_lane_utilisation"):
                    r["valu_lane_utilisation"] = c["valu_lane_utilisation"]
                if c.get("SQ_INSTS_SALU"):  # one scalar ALU per CU, one instruction per cycle
                    r["salu_issue_frac"] = c["SQ_INSTS_SALU"] / (cu_count * SHADER_CLOCK_HZ * k_s)
            r["counters_stale"] = False
            r["counters_from"] = c.get("source", "profiles/r03_counters.json")
        return r

    total_samples = args.width * args.height * args.spp
    m = measure(scene, cam, st, args.steps, args.warmup, args.balance)
    if args.depth == 0:
        assert m["seg_total"] == total_samples, (m["seg_total"], total_samples)
    def paths_kernel(scene_, spp_, depth_):
        """The fused path kernel the launcher picks (kernels.hip, launch_render_tiles): the pooled form for scenes whose traversal
        arrays exceed 1 MB, at least two passes of 8 samples and two segments; otherwise one pass per walk."""
        i = scene_.object.info()
        big = i.inner_count * 256 + i.packet_count * 384 > (1 << 20)
        return "render_paths_pooled_kernel" if (big and depth_ >= 2 and spp_ >= 16) else "render_paths_kernel"

    kname = paths_kernel(scene, args.spp, args.depth) if args.depth > 0 else ("render_tiles_packet_kernel" if args.traversal == "packets" else "render_tiles_kernel")
    key = workload_key(scene_name, args.width, args.height, args.spp, args.tile, args.depth, args.traversal)
    cu = ctx.cu_count

    def secondary(scene_, cam_, name_, desc_, spp, depth, steps=3, warmup=1):
        st2 = mp.RenderSettings(args.tile, spp, (args.width, args.height), seed=args.seed, max_depth=depth)
        m2 = measure(scene_, cam_, st2, steps, warmup)
        samples = args.width * args.height * spp
        return {
            "workload": f"{desc_}; {args.width}x{args.height} {spp}spp tile{args.tile} seed{args.seed:#x} "
                        + ("depth1 (reference semantics)" if depth == 0 else f"MP_FLAG_PATHS max_depth {depth} (build-defined extension: grey diffuse 0.75, white sky; no reference counterpart; rays = traced segments)"),
            "value": m2["seg_total"] * steps / m2["elapsed"] / 1e6, "unit": "Mrays/s", "steps": steps, "warmup": warmup,
            "ms_per_step": m2["elapsed"] / steps * 1e3, "samples_per_s": samples * steps / m2["elapsed"],
            "segments_per_sample": m2["seg_total"] / samples,
            "roofline": roofline(paths_kernel(scene_, spp, depth) if depth > 0 else "render_tiles_packet_kernel", m2,
                                 B_RAY_BOUNCE if depth > 0 else B_RAY_DEPTH1,
                                 workload_key(name_, args.width, args.height, spp, args.tile, depth, "packets"), cu),
        }

    ext = {}
    if not args.no_extension and args.depth == 0 and args.scene == "atrium" and args.traversal == "packets":
        ext["paths_depth8"] = secondary(scene, cam, scene_name, scene_desc, 64, 8, steps=2, warmup=1)  # configs[2]
        tscene, tcam, tname, tdesc = make_scene(mp, ctx, TEAPOT, 1.0)
        ext["teapot_c2"] = {"depth1": secondary(tscene, tcam, tname, tdesc, 256, 0), "paths_depth8": secondary(tscene, tcam, tname, tdesc, 256, 8)}

    if rank == 0:
        b_ray = B_RAY_BOUNCE if args.depth > 0 else B_RAY_DEPTH1
        out = {
            "metric": "Mrays/s",
            "value": m["seg_total"] * args.steps / m["elapsed"] / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": m["elapsed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" + (" (REHEARSAL: ranks share one GPU over gloo; not a measurement)" if rehearsal else ""),
            "samples_per_s": total_samples * args.steps / m["elapsed"],
            "config": {
                "workload": f"{scene_desc}; {args.width}x{args.height} {args.spp}spp tile{args.tile} seed{args.seed:#x} "
                            + ("depth1 (reference semantics: primary ray + |d.n|, worker.rs:51-66)" if args.depth == 0 else
                               f"paths max_depth {args.depth} (build-defined extension, no reference counterpart; rays = traced segments)"),
                "rays_per_step": m["seg_total"],
                "parallelism": f"tiles round-robin over {world} rank(s)" + (" + RCCL gather to rank 0" if world > 1 else "")
                               + (f"; {args.progressive} progressive passes per frame, accumulators resident on their GPUs, one gather per frame" if args.progressive > 1 else ""),
            },
            "roofline": roofline(kname, m, b_ray, key, cu),
        }
        out.update(ext)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, mp)
        if args.check and m["img"] is not None:
            import numpy as np

            from oracle import pyoracle as po

            ob, ocam = oracle_scene(po, mp, args.scene, args.detail)
            s = po.build_sampler(ocam, args.width, args.height)
            host = m["img"].cpu().numpy()
            bad = 0
            tiles = m["tiles"]
            for t in tiles[:: max(1, len(tiles) // 6)]:
                if args.depth > 0:
                    f, _, _ = ob.render_tile_paths(s, args.width, args.height, args.spp, args.seed, args.depth, t.min_x, t.min_y, t.max_x, t.max_y)
                else:
                    f, _ = ob.render_tile(s, args.width, args.height, args.spp, args.seed, t.min_x, t.min_y, t.max_x, t.max_y)
                bad += int(np.sum(host[t.min_y:t.max_y, t.min_x:t.max_x].view(np.uint32) != f.view(np.uint32)))
            out["check_mismatches"] = bad
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
