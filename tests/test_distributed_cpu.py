"""N>1 path on CPU: world_size-2 (and 3) gloo processes exercise the tile partition, the padded gather and the
rank-major reassembly of minipath_amd.distributed with a synthetic per-pixel pattern in place of rendered tiles."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as tmp

from tests.conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pattern(x, y):
    return np.stack([x * 1.0, y * 1.0, x * 1000.0 + y, np.ones_like(x, dtype=np.float64)], -1).astype(np.float32)


def _worker(rank, world, port, res, ts, out_path):
    sys.path.insert(0, ROOT)
    from minipath_amd.distributed import gather_shards, plan_shards
    from minipath_amd.screen_block import ScreenBlock, tile_ordering

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h = res
    tiles = tile_ordering(ScreenBlock(0, 0, w, h), ts)
    plan = plan_shards(tiles, world)
    mine = plan.shards[rank]
    shard = torch.full((plan.per_rank, ts, ts, 4), -1.0, dtype=torch.float32)
    for i, t in enumerate(mine):  # "render": tile-major, row stride ts, clipped tiles leave the rest untouched
        ys, xs = np.mgrid[t.min_y:t.max_y, t.min_x:t.max_x]
        shard[i, : t.height(), : t.width()] = torch.from_numpy(_pattern(xs, ys))
    # progressive shape ("accumulators stay sharded; gather only per displayed pass / at end"): three "passes" add to
    # the rank's own shard; nothing is exchanged until the one gather below
    for k in (1, 2, 3):
        for i, t in enumerate(mine):
            shard[i, : t.height(), : t.width(), 3] += float(k)
    cat = gather_shards(shard, plan, rank)
    if rank == 0:
        order = plan.gather_order
        assert cat.shape[0] == world * plan.per_rank == len(order)
        img = np.full((h, w, 4), np.nan, np.float32)
        seen = np.zeros((h, w), np.int32)
        for slot in plan.keep_indices():  # the un-tile (mp_untile on the GPU)
            t = order[slot]
            img[t.min_y:t.max_y, t.min_x:t.max_x] = cat[slot, : t.height(), : t.width()].numpy()
            seen[t.min_y:t.max_y, t.min_x:t.max_x] += 1
        ys, xs = np.mgrid[0:h, 0:w]
        want = _pattern(xs, ys)
        want[..., 3] += 6.0  # the three passes every rank accumulated locally
        ok = bool(np.array_equal(img, want) and np.all(seen == 1))
        np.save(out_path, np.array([ok, len(tiles), plan.per_rank], dtype=np.int64))
    else:
        assert cat is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,res,ts", [(2, (200, 120), 64), (3, (130, 70), 32), (2, (64, 64), 64)])
def test_gloo_shard_gather_reassemble(tmp_path, world, res, ts):
    out = str(tmp_path / "result.npy")
    tmp.spawn(_worker, args=(world, _free_port(), res, ts, out), nprocs=world, join=True)
    ok, ntiles, per_rank = np.load(out)
    assert ok == 1
    assert per_rank == -(-ntiles // world)


def test_plan_shards_properties():
    from minipath_amd.distributed import plan_shards
    from minipath_amd.screen_block import ScreenBlock, tile_ordering

    tiles = tile_ordering(ScreenBlock(0, 0, 1920, 1080), 64)
    for world in (1, 2, 4, 8):
        plan = plan_shards(tiles, world)
        flat = [t for s in plan.shards for t in s]
        assert sorted(t.as_struct().as_tuple() for t in flat) == sorted(t.as_struct().as_tuple() for t in tiles)
        assert max(len(s) for s in plan.shards) - min(len(s) for s in plan.shards) <= 1
        assert len(plan.gather_order) == world * plan.per_rank
        assert [plan.gather_order[i] for i in plan.keep_indices()] == flat
    with pytest.raises(ValueError):
        plan_shards(tiles, 0)
