"""Multi-GPU rendering: one process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm).

screen_block tiles are the unit of work sharing in the reference (threads pull tiles from one queue,
machinery.rs:74-105); here ranks own a static round-robin share of the row-major tile grid and every sample of a
pixel stays on one device, so the per-pixel accumulation order -- and therefore the image -- is independent of the
number of GPUs.  There is no collective on the data path; the only exchange is the final gather of the tile-major
framebuffer shards to rank 0, followed by the un-tile kernel there.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

from .screen_block import ScreenBlock

EMPTY = ScreenBlock(0, 0, 0, 0)


@dataclass(frozen=True)
class ShardPlan:
    """Static partition of a tile list over `world` ranks."""

    world: int
    per_rank: int                      # tiles per shard, padded to equal size for the gather
    shards: tuple                      # shards[r] = tiles of rank r (unpadded)

    @property
    def gather_order(self) -> List[ScreenBlock]:
        """Tile of every slot of the rank-major concatenation of the padded shards (EMPTY = padding)."""
        out: List[ScreenBlock] = []
        for tl in self.shards:
            out += list(tl) + [EMPTY] * (self.per_rank - len(tl))
        return out

    def keep_indices(self) -> List[int]:
        return [i for i, t in enumerate(self.gather_order) if not t.is_empty()]


def plan_shards(tiles: Sequence[ScreenBlock], world: int) -> ShardPlan:
    if world < 1:
        raise ValueError("world must be >= 1")
    shards = tuple(tuple(tiles[r::world]) for r in range(world))
    per_rank = max((len(s) for s in shards), default=0)
    return ShardPlan(world, per_rank, shards)


def gather_shards(shard, plan: ShardPlan, rank: int, group=None, dst: int = 0):
    """Gather equal-size tile-major shards [per_rank, ts, ts, 4] to `dst`.  Returns the rank-major concatenation
    [world*per_rank, ts, ts, 4] on dst, None elsewhere.  With RCCL this is grouped send/recv: every peer uses its own
    xGMI link to the root."""
    import torch
    import torch.distributed as dist

    if plan.world == 1:
        return shard
    assert shard.shape[0] == plan.per_rank
    bufs = [torch.empty_like(shard) for _ in range(plan.world)] if rank == dst else None
    dist.gather(shard, bufs, dst=dst, group=group)
    return torch.cat(bufs, 0) if rank == dst else None


class DistributedFrame:
    """Per-rank driver used by bench.py: render my shard in one launch straight into the gather source buffer, gather
    into one preallocated rank-major buffer on rank 0, un-tile there (padding slots are empty blocks the kernel skips).
    Nothing is allocated per frame.

    With more than one rank the frames are pipelined one deep: the gather of frame k runs on RCCL's stream while the compute
    stream already renders frame k+1 into the second shard buffer, so the xGMI transfer is off the critical path.  step()
    therefore returns the image of the PREVIOUS frame (None for the first one); flush() completes the last frame."""

    def __init__(self, scene, camera, settings, rank: int, world: int, tiles: Optional[Sequence[ScreenBlock]] = None):
        import torch

        from . import _lib
        from .renderer import FrameRenderer
        from .screen_block import tile_ordering

        w, h = settings.resolution
        self.rank, self.world, self.settings = rank, world, settings
        self._camera = camera
        self.all_tiles = list(tiles) if tiles is not None else tile_ordering(ScreenBlock(0, 0, w, h), settings.tile_size)
        self.plan = plan_shards(self.all_tiles, world)
        ts = settings.tile_size
        dev = torch.device("cuda", scene.object.ctx.device_id)
        nbuf = 2 if world > 1 else 1
        self._shards = [torch.zeros((max(self.plan.per_rank, 1), ts, ts, 4), dtype=torch.float32, device=dev) for _ in range(nbuf)]
        self.renderer = FrameRenderer(scene, camera, settings, tiles=self.plan.shards[rank], tile_buf=self._shards[0])
        self._gather_buf = self._gather_views = self._order = None
        self._pending = None   # (work, want_u8) of the gather in flight
        self._frame = 0
        if rank == 0 and world > 1:
            self._gather_buf = torch.zeros((world * self.plan.per_rank, ts, ts, 4), dtype=torch.float32, device=dev)
            self._gather_views = [self._gather_buf[r * self.plan.per_rank:(r + 1) * self.plan.per_rank] for r in range(world)]
            order = self.plan.gather_order
            self._order = (order, (_lib.Block * max(len(order), 1))(*[t.as_struct() for t in order]))

    @property
    def shard(self):
        """The shard buffer the next render writes."""
        return self._shards[self._frame % len(self._shards)]

    @property
    def rays_per_frame_local(self) -> int:
        return self.renderer.rays_per_frame

    def render_local(self):
        self.renderer.tile_buf = self.shard
        return self.renderer.render()

    def rebalance(self):
        """Profile-guided tile hand-out for this rank's following launches (FrameRenderer.rebalance): purely local, the shard
        layout and the gather do not change."""
        return self.renderer.rebalance()

    def repartition_by_cost(self):
        """Cost-balanced partition for the following frames (opt-in; bench.py --balance lpt): every rank contributes the shader
        cycles its launches measured per tile (FrameRenderer.tile_cost), all ranks compute the SAME longest-processing-time
        assignment from the all-gathered costs (tiles by descending cost, each to the least loaded rank, ties to the lower
        rank / tile index), and rebuild their shard: tile list (already expensive-first, so no hand-out permutation is needed),
        shard and gather buffers, un-tile order.  The image does not depend on the partition.  Collective: call on every rank,
        with no frame in flight (after flush())."""
        import torch
        import torch.distributed as dist

        from . import _lib
        from .renderer import FrameRenderer

        assert self._pending is None, "flush() before repartitioning"
        world, per = self.world, max(self.plan.per_rank, 1)
        mine = self.plan.shards[self.rank]
        dev = self.renderer.device
        local = torch.zeros(per, dtype=torch.int64, device=dev)
        local[: len(mine)] = self.renderer.tile_cost[: len(mine)]
        if world > 1:
            gathered = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
        else:
            gathered = [local]
        costs = [g.cpu().tolist() for g in gathered]
        index = {t: i for i, t in enumerate(self.all_tiles)}
        cost = [0] * len(self.all_tiles)
        for r, shard in enumerate(self.plan.shards):
            for k, t in enumerate(shard):
                cost[index[t]] = int(costs[r][k])
        order = sorted(range(len(cost)), key=lambda i: (-cost[i], i))
        loads, shards = [0] * world, [[] for _ in range(world)]
        for i in order:
            r = min(range(world), key=lambda q: (loads[q], q))
            shards[r].append(self.all_tiles[i])
            loads[r] += cost[i]
        self.plan = ShardPlan(world, max(len(s) for s in shards), tuple(tuple(s) for s in shards))
        ts = self.settings.tile_size
        self._shards = [torch.zeros((max(self.plan.per_rank, 1), ts, ts, 4), dtype=torch.float32, device=dev) for _ in range(len(self._shards))]
        old = self.renderer
        self.renderer = FrameRenderer(old.scene, self._camera, self.settings, tiles=self.plan.shards[self.rank], tile_buf=self._shards[0])
        self._frame = 0
        if self.rank == 0 and world > 1:
            self._gather_buf = torch.zeros((world * self.plan.per_rank, ts, ts, 4), dtype=torch.float32, device=dev)
            self._gather_views = [self._gather_buf[r * self.plan.per_rank:(r + 1) * self.plan.per_rank] for r in range(world)]
            o = self.plan.gather_order
            self._order = (o, (_lib.Block * max(len(o), 1))(*[t.as_struct() for t in o]))
        return loads

    def _complete(self):
        """Wait (on the current stream) for the gather in flight and un-tile its frame on rank 0."""
        if self._pending is None:
            return None, None
        work, want_u8 = self._pending
        self._pending = None
        work.wait()
        if self.rank != 0:
            return None, None
        return self.renderer.untile(self._gather_buf, self._order, want_u8=want_u8, reuse=True)

    def step(self, want_u8: bool = True, kernel_events=None):
        """One frame.  world == 1: returns (image f32, image u8) of this frame.  world > 1: launches this frame's render,
        completes the PREVIOUS frame (gather wait + un-tile; returns its images on rank 0, (None, None) elsewhere and for
        the first frame), then starts this frame's gather.  kernel_events = (start, end) torch.cuda.Event pair recorded around
        the render launch on the current stream."""
        import torch.distributed as dist

        if kernel_events is not None:
            kernel_events[0].record()
        src = self.render_local()
        if kernel_events is not None:
            kernel_events[1].record()
        self._frame += 1
        if self.world == 1:
            return self.renderer.untile(want_u8=want_u8, reuse=True)
        prev = self._complete()
        work = dist.gather(src, self._gather_views, dst=0, async_op=True)
        self._pending = (work, want_u8)
        return prev

    def flush(self):
        """Complete the frame whose gather is still in flight (world > 1); returns its images on rank 0."""
        return self._complete()

    # ---- progressive rendering (BASELINE configs[4]; SURVEY 8e: "accumulators stay sharded; gather only per displayed pass /
    # at end") --------------------------------------------------------------------------------------------------------------
    def render_pass(self, begin: int, count: int = 0) -> int:
        """Adds samples [begin, begin+count) (count 0 = through the last) to the running per-pixel state of THIS rank's shard
        (FrameRenderer.render_pass); purely local: nothing is exchanged.  Returns the next sample index.  All passes of one
        progressive render go to the same shard buffer (no frame pipelining: call with no gather in flight)."""
        assert self._pending is None, "flush() before a progressive pass"
        self.renderer.tile_buf = self._shards[0]
        self._frame = 0
        return self.renderer.render_pass(begin, count)

    def gather_image(self, samples_done: int, want_u8: bool = True):
        """Collective: gathers the shards to rank 0 and un-tiles them there -- the finished frame if samples_done ==
        sample_count (the last pass wrote the means), otherwise a preview of the samples drawn so far (mp_untile_preview); the
        shards keep their running state either way.  Returns (f32 image, u8 image) on rank 0, (None, None) elsewhere."""
        import torch.distributed as dist

        total = int(self.settings.sample_count)
        preview = None if samples_done >= total else int(samples_done)
        src = self._shards[0]
        if self.world == 1:
            return self.renderer.untile(src, want_u8=want_u8, reuse=True, preview_samples=preview)
        dist.gather(src, self._gather_views, dst=0)
        if self.rank != 0:
            return None, None
        return self.renderer.untile(self._gather_buf, self._order, want_u8=want_u8, reuse=True, preview_samples=preview)
