"""Generates tests/golden/configs_golden.npz: golden pixel blocks of every BASELINE.json config AT ITS OWN SIZE, from the CPU oracle.

The Rust reference cannot run in this pipeline (no cargo/rustc), so these are NOT reference outputs: they freeze the oracle's
restatement at the configured resolutions / sample counts / depths, so that the GPU suite exercises every config at full size
(`tests/test_gpu_configs.py`) without running the oracle on the GPU box for minutes.  Blocks are small rectangles (centre,
silhouette, clipped bottom-row tile) rendered with ALL samples of the config.

  C2  teapot.obj 1920x1080 256 spp, depth 1 (reference semantics) and max depth 8 (build-defined extension)
  C3  Sponza stand-in (minipath_amd.scenes.atrium(1), full detail, built by the ORACLE's own restated builder) 1920x1080 64 spp,
      depth 1 and max depth 8
  C4  stand-in 3840x2160 1024 spp, depth 1 and 8: blocks inside tiles of rank 3's shard of the 8-rank round-robin partition
  C5  stand-in 3840x2160 65 536 spp, max depth 16, chunked accumulation rule (MP_FLAG_CHUNKED_SUM)

Run from the repo root (about 5 minutes on 8 cores):  python tests/golden/make_config_golden.py
"""
import ctypes as C
import hashlib
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from minipath_amd import scenes  # noqa: E402  (numpy-only scene generator; no product code runs here)

SEED = 0x5EED
HERE = os.path.dirname(os.path.abspath(__file__))


def atrium_camera():
    cam = po.Camera()
    po.lib().mpo_camera_default(C.byref(cam))
    eye, at, fnum = scenes.ATRIUM_VIEW
    po.lib().mpo_camera_look_at(C.byref(cam), po.vec3(*eye), po.vec3(*at), po.vec3(0, 1, 0))
    cam.f_number = fnum
    return cam


def render_blocks(bvh, cam, res, spp, depth, blocks, chunked=False):
    """Every block on its own thread (the oracle releases the GIL inside ctypes calls)."""
    s = po.build_sampler(cam, *res)
    out = [None] * len(blocks)

    def work(k):
        b = blocks[k]
        if depth:
            f, _, _ = bvh.render_tile_paths(s, res[0], res[1], spp, SEED, depth, *b)
        else:
            f, _ = bvh.render_tile(s, res[0], res[1], spp, SEED, *b)
        out[k] = f

    po.lib().mpo_set_chunked_sum(1 if chunked else 0)
    try:
        th = [threading.Thread(target=work, args=(k,)) for k in range(len(blocks))]
        [t.start() for t in th]
        [t.join() for t in th]
    finally:
        po.lib().mpo_set_chunked_sum(0)
    return out


def main():
    t0 = time.time()
    g = {}
    cases = []

    def add(name, scene, res, spp, depth, blocks, imgs, chunked=False):
        cases.append(name)
        g[f"{name}_meta"] = np.array([res[0], res[1], spp, depth, 1 if chunked else 0, 1 if scene == "atrium" else 0], np.uint32)
        g[f"{name}_blocks"] = np.array(blocks, np.uint32)
        for k, f in enumerate(imgs):
            g[f"{name}_b{k}"] = f.view(np.uint32)
        hits = float(np.mean([(f[..., 3] > 0).mean() for f in imgs]))
        print(f"{name}: {len(blocks)} blocks, mean alpha coverage {hits:.2f}, mean grey {np.mean([f[..., 0].mean() for f in imgs]):.4f}  [{time.time() - t0:.0f} s]", flush=True)

    # ---- C2: teapot -------------------------------------------------------------------------------------------------
    tb = po.Bvh.from_obj(os.path.join(HERE, "teapot.obj"))
    tcam = po.teapot_camera()
    res = (1920, 1080)
    # body centre / spout silhouette against the background / clipped bottom-row tile (rows 1024..1080 are a 56-pixel tile)
    blocks = [(944, 600, 976, 616), (1290, 470, 1322, 486), (928, 1040, 960, 1056)]
    for depth in (0, 8):
        add(f"c2_d{depth}", "teapot", res, 256, depth, blocks, render_blocks(tb, tcam, res, 256, depth, blocks))

    # ---- C3..C5: the Sponza stand-in at full detail, oracle's own builder ------------------------------------------------
    pos, nrm, tex, tri = scenes.atrium(1, 1.0)
    ab = po.Bvh.build(pos, nrm, tex, tri)
    print(f"atrium built by the oracle: {ab.n_inner} inner, {ab.n_packets} packets, depth {ab.depth}  [{time.time() - t0:.0f} s]", flush=True)
    g["atrium_counts"] = np.array([ab.n_inner, ab.n_packets, ab.n_vertices, ab.depth, ab.root, tri.shape[0]], np.uint32)
    g["atrium_sha256"] = np.array([hashlib.sha256(ab.inner_nodes_bytes().tobytes()).hexdigest(),
                                   hashlib.sha256(ab.packets_bytes().tobytes()).hexdigest(),
                                   hashlib.sha256(ab.tri_shading().tobytes()).hexdigest()])
    acam = atrium_camera()
    blocks = [(952, 536, 968, 552), (300, 200, 316, 216), (1700, 1050, 1716, 1066)]
    for depth in (0, 8):
        add(f"c3_d{depth}", "atrium", res, 64, depth, blocks, render_blocks(ab, acam, res, 64, depth, blocks))
    # C4: rank 3 of 8 owns tiles 3::8 of the row-major 60x34 grid of the 4K frame
    res4 = (3840, 2160)
    tiles = po.tile_ordering(0, 0, res4[0], res4[1], 64)
    shard = tiles[3::8]
    picks = [shard[10], shard[len(shard) // 2], shard[-1]]  # the last one lies in the clipped bottom row (48 pixels high)
    blocks = [(int(t[0]) + 20, int(t[1]) + 12, int(t[0]) + 28, int(t[1]) + 20) for t in picks]
    g["c4_shard_rank_world"] = np.array([3, 8], np.uint32)
    for depth in (0, 8):
        add(f"c4_d{depth}", "atrium", res4, 1024, depth, blocks, render_blocks(ab, acam, res4, 1024, depth, blocks))
    # C5: 65 536 spp, depth 16, chunked accumulation; two 4x4 blocks of one tile
    t5 = tiles[34 * 60 // 2 + 31]  # a tile in the middle of the frame
    blocks = [(int(t5[0]) + 8, int(t5[1]) + 8, int(t5[0]) + 12, int(t5[1]) + 12), (int(t5[0]) + 50, int(t5[1]) + 40, int(t5[0]) + 54, int(t5[1]) + 44)]
    g["c5_tile"] = np.array(t5, np.uint32)
    add("c5_d16", "atrium", res4, 65536, 16, blocks, render_blocks(ab, acam, res4, 65536, 16, blocks, chunked=True), chunked=True)
    g["cases"] = np.array(cases)
    g["seed"] = np.uint64(SEED)
    np.savez_compressed(os.path.join(HERE, "configs_golden.npz"), **g)
    print("wrote configs_golden.npz", os.path.getsize(os.path.join(HERE, "configs_golden.npz")), "bytes", f"[{time.time() - t0:.0f} s]")


if __name__ == "__main__":
    main()
