#!/usr/bin/env python3
"""bench.py -- Mrays/s of the per-pixel sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one frame of synthetic input: every pixel x every sample of the workload is
generated, traced, shaded and accumulated on the GPU (one launch per rank), the per-rank tile shards are gathered
to rank 0 over RCCL (N > 1) and scattered into the image-major framebuffer.  At N > 1 the frames are pipelined one deep
(the gather of frame k runs while frame k+1 renders); every frame is complete -- gathered and un-tiled -- before the closing
barrier of the timed region.  Scene (BVH) and camera are resident in
HBM before the timed region; the framebuffer stays in HBM.

Workload at N = 1 (BASELINE.json configs[1]): teapot.obj, 1920x1080, 256 spp, tile 64, seed 0x5EED, teapot view of
benches/render_teapot.rs:12-19.  `value` is measured with the REFERENCE SEMANTICS (primary ray + |d.n| shading = depth 1:
the reference has no bounce loop), the only mode whose results can be identical to the reference's and whose CPU
path is the reference's algorithm; a "ray" is one Object::intersect call, so rays == samples there.  configs[1] also says
"max depth 8": that exists only as this build's path extension (MP_FLAG_PATHS); the same frame with
--depth 8 semantics is measured in the same run and reported under "paths_depth8" (rays = traced path segments).

Multi-GPU (--gpus N, launched by torch.distributed.run): tiles are sharded round-robin over the ranks, the total
work is fixed ("strong" scaling), no collective on the data path except the final framebuffer gather.

After the warmup frames every rank orders the hand-out of ITS tiles to its waves by the cost the warmup measured (expensive
tiles first; mp_launch_extras / FrameRenderer.rebalance).  The work and the image are unchanged (bit-identical, tested); the
tail of each launch gets shorter, which matters most for the small per-rank launches at N > 1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_RAY_DEPTH1 = 136  # algorithmic HBM bytes per ray segment of the depth-1 wavefront formulation
B_RAY_BOUNCE = 320  # compacted wavefront bounce mode (build-defined path extension, --depth N)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--scene", default=os.path.join(ROOT, "tests", "golden", "teapot.obj"))
    ap.add_argument("--detail", type=float, default=1.0, help="atrium tessellation (1.0 = 258 k triangles)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tile-stride", type=int, default=11, help="cpu_baseline renders every k-th tile")
    ap.add_argument("--cpu-threads", type=int, default=16, help="upper bound on cpu_baseline worker threads")
    ap.add_argument("--traversal", default="packets", choices=["packets", "groups"])
    ap.add_argument("--depth", type=int, default=0, help="0 = reference semantics (default); N >= 1 = build-defined path extension with at most N segments")
    ap.add_argument("--balance", default="static", choices=["static", "lpt"],
                    help="N > 1: 'lpt' re-partitions the tiles over the ranks by the cost the warmup frames measured (opt-in)")
    ap.add_argument("--wavefront", action="store_true", help="with --depth N: staged evaluation, bounce rays sorted into packets")
    ap.add_argument("--no-extension", action="store_true", help="skip the extra 'paths_depth8' measurement")
    ap.add_argument("--check", action="store_true", help="compare a few tiles of the GPU frame with the oracle")
    return ap.parse_args()


def cpu_baseline(args):
    """The oracle (C restatement of the reference CPU path: one thread per core, atomic tile queue,
    machinery.rs:31-116) timed on this box's host cores on a bounded sample of the SAME workload."""
    from oracle import pyoracle as po

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a one-GPU box's CPU share is 16 cores however many the host exposes; `cores` = the threads actually used
    cores = min(cores, args.cpu_threads)
    if args.scene == "atrium":
        import ctypes as C

        from minipath_amd import scenes

        b = po.Bvh.build(*scenes.atrium(1, args.detail))
        cam = po.Camera()
        po.lib().mpo_camera_default(C.byref(cam))
        po.lib().mpo_camera_look_at(C.byref(cam), po.vec3(-16.0, 4.2, 0.8), po.vec3(12.0, 5.5, -0.5), po.vec3(0, 1, 0))
        cam.f_number = 4.0
        s = po.build_sampler(cam, args.width, args.height)
    else:
        b = po.Bvh.from_obj(args.scene)
        s = po.build_sampler(po.teapot_camera(), args.width, args.height)
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
        "sample": f"every {stride}th of the {ntiles} 64x64 tiles of the same frame at full {args.spp} spp "
                  f"({rays / 1e6:.1f} Mrays, {secs:.1f} s wall on {cores} threads); C restatement of the reference CPU path, "
                  "not the Rust reference (no Rust toolchain in this pipeline)",
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import minipath_amd as mp

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
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
    if args.scene == "atrium":  # Sponza stand-in (BASELINE configs[2..4]; data/Sponza is an empty submodule)
        from minipath_amd import scenes

        scene = mp.Scene(mp.TriangleBvh.build(*scenes.atrium(1, args.detail), ctx))
        cam = scenes.atrium_camera()
    else:
        scene = mp.Scene(mp.TriangleBvh.with_obj(args.scene, ctx))
        cam = mp.Camera.teapot_view()
    st = mp.RenderSettings(args.tile, args.spp, (args.width, args.height), seed=args.seed, traversal=args.traversal, max_depth=args.depth,
                           wavefront=args.wavefront)
    frame = DistributedFrame(scene, cam, st, rank, world)
    all_tiles = frame.all_tiles
    total_rays = args.width * args.height * args.spp
    events = []

    def step(timed):
        """Render my shard (one launch), gather to rank 0 over RCCL, un-tile there."""
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        img, _ = frame.step(want_u8=True, kernel_events=ev)
        if timed:
            events.append(ev)
        return img

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    frame.flush()
    if args.warmup > 0 and args.balance == "lpt" and world > 1:
        frame.repartition_by_cost()  # cost-balanced shards, expensive tiles first (collective)
    elif args.warmup > 0:
        frame.rebalance()  # hand the tiles the warmup frames found expensive to the waves first (same image, shorter tail)
    barrier()
    t0 = time.perf_counter()
    img = None
    for _ in range(args.steps):
        out_img = step(True)  # N > 1: the previous frame's image (the gather of frame k overlaps the render of frame k+1)
        img = out_img if out_img is not None else img
    last, _ = frame.flush()   # completes the last frame's gather + un-tile inside the timed region
    img = last if last is not None else img
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in events]
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        kmax = torch.tensor([sum(kernel_ms) / max(len(kernel_ms), 1)], dtype=torch.float64, device=dev)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        k_ms = float(kmax.item())
    else:
        k_ms = sum(kernel_ms) / max(len(kernel_ms), 1)

    # the same frame with the build-defined max-depth-8 path extension (configs[1] "max depth 8"), outside the timed region above
    ext = None
    if args.depth == 0 and not args.no_extension:
        st8 = mp.RenderSettings(args.tile, args.spp, (args.width, args.height), seed=args.seed, max_depth=8)
        frame8 = DistributedFrame(scene, cam, st8, rank, world)
        ev8 = []
        for i in range(4):  # 1 warmup + 3 timed
            if i == 1:
                barrier()
                t8 = time.perf_counter()
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            frame8.step(want_u8=True, kernel_events=ev)
            if i == 0:
                frame8.flush()
                frame8.rebalance()
            if i >= 1:
                ev8.append(ev)
        frame8.flush()
        barrier()
        el8 = time.perf_counter() - t8
        seg8 = torch.tensor([int(frame8.renderer.segments.item())], dtype=torch.int64, device=dev)
        k8 = torch.tensor([sum(a.elapsed_time(b) for a, b in ev8) / 3.0, el8], dtype=torch.float64, device=dev)
        seg8_local = int(seg8.item())
        if world > 1:
            dist.all_reduce(seg8, op=dist.ReduceOp.SUM)
            dist.all_reduce(k8, op=dist.ReduceOp.MAX)
        seg8_total, k8_ms, el8 = int(seg8.item()), float(k8[0].item()), float(k8[1].item())
        ach8 = seg8_local * B_RAY_BOUNCE / (k8_ms * 1e-3) / 1e9
        ext = {
            "workload": "same frame, MP_FLAG_PATHS max_depth 8 (build-defined extension: diffuse 0.75, white sky; no reference counterpart)",
            "value": seg8_total * 3 / el8 / 1e6, "unit": "Mrays/s (traced path segments)", "steps": 3,
            "ms_per_step": el8 / 3 * 1e3, "samples_per_s": args.width * args.height * args.spp * 3 / el8,
            "segments_per_sample": seg8_total / (args.width * args.height * args.spp),
            "roofline": {"bound": "hbm", "kernel": "render_paths_kernel", "achieved": ach8, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach8 / HBM_PEAK_GBS, "kernel_ms": k8_ms, "bytes_per_ray": B_RAY_BOUNCE, "traffic": None},
        }

    # rays = Object::intersect calls: W*H*spp for the reference semantics, traced path segments for --depth N
    seg_local = int(frame.renderer.segments.item())
    if world > 1:
        segt = torch.tensor([seg_local], dtype=torch.int64, device=dev)
        dist.all_reduce(segt, op=dist.ReduceOp.SUM)
        seg_total = int(segt.item())
    else:
        seg_total = seg_local
    if args.depth == 0:
        assert seg_total == total_rays, (seg_total, total_rays)
    total_samples = total_rays
    total_rays = seg_total
    b_ray = B_RAY_BOUNCE if args.depth > 0 else B_RAY_DEPTH1

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rays * args.steps / elapsed / 1e6
        # dominant kernel: render_tiles_kernel.  Algorithmic bytes per launch = B_ray x rays of this rank's launch
        # (DESIGN.md "Roofline"); duration from HIP events on the launch stream.
        rays_per_launch = seg_local
        achieved = rays_per_launch * b_ray / (k_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        default_workload = (args.scene.endswith("teapot.obj") and (args.width, args.height, args.spp, args.tile) == (1920, 1080, 256, 64)
                            and args.depth == 0 and args.traversal == "packets")
        if world == 1 and default_workload and os.path.exists(tp):  # measured for the one-GPU launch of the default workload only
            try:
                traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s",
            "value": value,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" + (" (REHEARSAL: ranks share one GPU over gloo; not a measurement)" if rehearsal else ""),
            "samples_per_s": total_samples * args.steps / elapsed,
            "config": {
                "workload": f"{os.path.basename(args.scene)} {args.width}x{args.height} {args.spp}spp tile{args.tile} "
                            f"seed{args.seed:#x} " + ("depth1 (reference semantics: primary ray + |d.n|, worker.rs:51-66)" if args.depth == 0 else
                                                     f"paths max_depth {args.depth} (build-defined extension, no reference counterpart; rays = traced segments)"),
                "rays_per_step": total_rays,
                "parallelism": f"tiles round-robin over {world} rank(s)" + (" + RCCL gather to rank 0" if world > 1 else ""),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "render_paths_kernel" if args.depth > 0 else ("render_tiles_packet_kernel" if args.traversal == "packets" else "render_tiles_kernel"),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel_ms": k_ms,
                "algorithmic_bytes_per_launch": rays_per_launch * b_ray,
                "bytes_per_ray": b_ray,
            },
        }
        if ext is not None:
            out["paths_depth8"] = ext
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        if args.check and img is not None:
            from oracle import pyoracle as po
            import numpy as np

            if args.scene == "atrium":
                import ctypes as C

                from minipath_amd import scenes

                ob = po.Bvh.build(*scenes.atrium(1, args.detail))
                ocam = po.Camera()
                po.lib().mpo_camera_default(C.byref(ocam))
                po.lib().mpo_camera_look_at(C.byref(ocam), po.vec3(-16.0, 4.2, 0.8), po.vec3(12.0, 5.5, -0.5), po.vec3(0, 1, 0))
                ocam.f_number = 4.0
                s = po.build_sampler(ocam, args.width, args.height)
            else:
                ob = po.Bvh.from_obj(args.scene)
                s = po.build_sampler(po.teapot_camera(), args.width, args.height)
            host = img.cpu().numpy()
            bad = 0
            for t in all_tiles[:: max(1, len(all_tiles) // 6)]:
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
