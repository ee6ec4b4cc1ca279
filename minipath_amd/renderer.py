"""Mirror of src/renderer: RenderSettings (mod.rs:7-13), render() and RenderProgress (machinery.rs:20-178),
Worker::render_tile (worker.rs:32-49)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib
from .camera import Camera, CameraSampler
from .scene import Context, Scene
from .screen_block import ScreenBlock, tile_ordering


@dataclass(frozen=True)
class RenderSettings:
    """renderer/mod.rs:7-13 plus the build-defined seed of the reproducible sample stream."""

    tile_size: int
    sample_count: int
    resolution: tuple
    seed: int = 0x5EED
    shuffle_tiles: bool = False
    traversal: str = "packets"  # "packets": 64 camera rays per wave share one BVH walk; "groups": 8 lanes per ray
    max_depth: int = 0  # 0: reference semantics (primary ray + |d.n|).  >= 1: build-defined path extension (MP_FLAG_PATHS)
    wavefront: bool = False  # with max_depth >= 1: staged evaluation, bounce rays sorted into packets (MP_FLAG_WAVEFRONT)
    chunked_sum: bool = False  # build-defined: f32 sums over 256-sample chunks, f64 total (MP_FLAG_CHUNKED_SUM; configs[4])
    image_u8_only: bool = False  # render(): keep only the reference's u8 image on the host (MP_FLAG_IMAGE_U8_ONLY; image_f32() then raises)

    def as_struct(self) -> _lib.SettingsStruct:
        if self.tile_size <= 0 or self.sample_count <= 0:
            raise ValueError("tile_size and sample_count are NonZeroU32")
        if self.traversal not in ("packets", "groups"):
            raise ValueError("traversal is 'packets' or 'groups'")
        return _lib.SettingsStruct(
            int(self.tile_size), int(self.sample_count), int(self.resolution[0]), int(self.resolution[1]),
            int(self.seed) & 0xFFFFFFFFFFFFFFFF,
            (_lib.MP_FLAG_SHUFFLE_TILES if self.shuffle_tiles else 0)
            | (_lib.MP_FLAG_TRAVERSAL_GROUPS if self.traversal == "groups" else 0)
            | (_lib.MP_FLAG_PATHS if self.max_depth > 0 else 0)
            | (_lib.MP_FLAG_WAVEFRONT if (self.wavefront and self.max_depth > 0) else 0)
            | (_lib.MP_FLAG_CHUNKED_SUM if self.chunked_sum else 0)
            | (_lib.MP_FLAG_IMAGE_U8_ONLY if self.image_u8_only else 0),
            int(self.max_depth),
            0,
            0,
        )


@dataclass(frozen=True)
class RenderProgressSnapshot:
    """machinery.rs:180-189."""

    finished: int
    total: int

    def percent(self) -> float:
        return 100.0 * self.finished / self.total


class RenderProgress:
    """machinery.rs:125-178."""

    def __init__(self, handle, settings: RenderSettings, keepalive):
        self._h = handle
        self._settings = settings
        self._keepalive = keepalive  # callbacks + scene must outlive the worker thread

    def progress(self) -> RenderProgressSnapshot:
        p = _lib.Progress()
        _lib.check(_lib.lib().mp_render_progress(self._h, C.byref(p)))
        return RenderProgressSnapshot(p.finished, p.total)

    def is_finished(self) -> bool:
        f = C.c_int()
        _lib.check(_lib.lib().mp_render_is_finished(self._h, C.byref(f)))
        return bool(f.value)

    def elapsed(self) -> float:
        ns = C.c_uint64()
        _lib.check(_lib.lib().mp_render_elapsed_ns(self._h, C.byref(ns)))
        return ns.value * 1e-9

    def abort(self) -> None:
        _lib.check(_lib.lib().mp_render_abort(self._h))

    def wait(self) -> None:
        _lib.check(_lib.lib().mp_render_wait(self._h))

    def image(self) -> np.ndarray:
        """RGBA u8 [h, w, 4] (copy taken under the image lock)."""
        w, h = self._settings.resolution
        out = np.zeros((h, w, 4), np.uint8)
        _lib.check(_lib.lib().mp_render_image_u8(self._h, out.ctypes.data))
        return out

    def image_f32(self) -> np.ndarray:
        """Pre-quantisation per-pixel means (worker.rs:44), [h, w, 4] f32."""
        w, h = self._settings.resolution
        out = np.zeros((h, w, 4), np.float32)
        _lib.check(_lib.lib().mp_render_image_f32(self._h, out.ctypes.data))
        return out

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().mp_render_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render(
    scene: Scene,
    camera: Camera,
    settings: RenderSettings,
    started_tile_callback: Optional[Callable[[ScreenBlock], None]] = None,
    finished_tile_callback: Optional[Callable[[ScreenBlock, RenderProgressSnapshot], None]] = None,
) -> RenderProgress:
    """render() (machinery.rs:20-123): asynchronous; callbacks run on the library's worker thread."""
    bvh = scene.object
    if bvh.ctx is None:
        raise _lib.MinipathError(1, "scene is host-only: build it with a Context")

    def _started(_user, blk):
        if started_tile_callback:
            started_tile_callback(ScreenBlock(*blk.as_tuple()))

    def _finished(_user, blk, prog):
        if finished_tile_callback:
            finished_tile_callback(ScreenBlock(*blk.as_tuple()), RenderProgressSnapshot(prog.finished, prog.total))

    # NULL when the caller passes no callback: the worker thread then never enters Python
    cb1 = _lib.STARTED_CB(_started) if started_tile_callback else _lib.STARTED_CB()
    cb2 = _lib.FINISHED_CB(_finished) if finished_tile_callback else _lib.FINISHED_CB()
    h = C.c_void_p()
    cam = camera._struct()
    st = settings.as_struct()
    _lib.check(_lib.lib().mp_render_begin(bvh.ctx.handle, bvh.handle, C.byref(cam), C.byref(st), cb1, cb2, None, C.byref(h)))
    return RenderProgress(h, settings, (cb1, cb2, scene))


def render_multi(
    scenes: Sequence[Scene],
    camera: Camera,
    settings: RenderSettings,
    started_tile_callback: Optional[Callable[[ScreenBlock], None]] = None,
    finished_tile_callback: Optional[Callable[[ScreenBlock, RenderProgressSnapshot], None]] = None,
) -> RenderProgress:
    """render() over several GPUs of this process (mp_render_begin_multi): scenes[i] is the same scene uploaded to its own
    Context; one library-owned worker thread per scene pulls tiles from the one shared queue (machinery.rs:51-116 with devices
    in place of cores).  Callbacks may run concurrently on several threads."""
    if not scenes:
        raise ValueError("no scenes")
    for sc in scenes:
        if sc.object.ctx is None:
            raise _lib.MinipathError(1, "scene is host-only: build it with a Context")

    def _started(_user, blk):
        if started_tile_callback:
            started_tile_callback(ScreenBlock(*blk.as_tuple()))

    def _finished(_user, blk, prog):
        if finished_tile_callback:
            finished_tile_callback(ScreenBlock(*blk.as_tuple()), RenderProgressSnapshot(prog.finished, prog.total))

    cb1 = _lib.STARTED_CB(_started) if started_tile_callback else _lib.STARTED_CB()
    cb2 = _lib.FINISHED_CB(_finished) if finished_tile_callback else _lib.FINISHED_CB()
    n = len(scenes)
    ctxs = (C.c_void_p * n)(*[sc.object.ctx.handle for sc in scenes])
    hs = (C.c_void_p * n)(*[sc.object.handle for sc in scenes])
    h = C.c_void_p()
    cam = camera._struct()
    st = settings.as_struct()
    _lib.check(_lib.lib().mp_render_begin_multi(ctxs, hs, n, C.byref(cam), C.byref(st), cb1, cb2, None, C.byref(h)))
    return RenderProgress(h, settings, (cb1, cb2, list(scenes)))


class MultiDeviceFrame:
    """One process, several GPUs, device-resident frame (mp_render_frame_multi): rank r = scenes[r] renders tiles r::n in one
    launch on its own device, shards travel to scenes[0]'s device by peer copies, un-tile there.  The single-process
    counterpart of minipath_amd.distributed.DistributedFrame (one process per GPU + RCCL)."""

    def __init__(self, scenes: Sequence[Scene], camera: Camera, settings: RenderSettings):
        import torch

        self.scenes, self.settings = list(scenes), settings
        n = len(self.scenes)
        self._ctxs = (C.c_void_p * n)(*[sc.object.ctx.handle for sc in self.scenes])
        self._hs = (C.c_void_p * n)(*[sc.object.handle for sc in self.scenes])
        self._sampler = camera.build_sampler(settings.resolution).as_struct()
        self._st = settings.as_struct()
        self.device = torch.device("cuda", self.scenes[0].object.ctx.device_id)
        w, h = settings.resolution
        self.image = torch.zeros((h, w, 4), dtype=torch.float32, device=self.device)
        self.image_u8 = torch.zeros((h, w, 4), dtype=torch.uint8, device=self.device)
        self.segments = 0

    def render(self):
        """One frame, asynchronous on the current torch stream of scenes[0]'s device; returns (f32 image, u8 image)."""
        import torch

        seg = C.c_uint64()
        _lib.check(
            _lib.lib().mp_render_frame_multi(
                self._ctxs, self._hs, len(self.scenes), C.byref(self._sampler), C.byref(self._st), self.image.data_ptr(),
                self.image_u8.data_ptr(), C.byref(seg), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream),
            )
        )
        self.segments = seg.value
        return self.image, self.image_u8

    def render_pass(self, begin: int, count: int = 0, gather: bool = True):
        """Progressive form (mp_render_pass_multi; BASELINE configs[4]): adds samples [begin, begin+count) of
        settings.sample_count (count 0 = through the last) to the running state of every rank's shard, which stays on its
        device.  gather=False: nothing leaves the devices; returns the next sample index.  gather=True: also gathers and
        un-tiles on scenes[0]'s device -- the finished frame after the last pass (bit-identical to render()), a preview
        (running sums / samples so far) before; returns (next sample index, f32 image, u8 image)."""
        import torch

        total = int(self.settings.sample_count)
        if not (0 <= begin < total) or count < 0 or begin + count > total:
            raise ValueError("pass outside [0, sample_count)")
        st = _lib.SettingsStruct.from_buffer_copy(self._st)
        st.flags |= _lib.MP_FLAG_ACCUMULATE
        st.pass_begin, st.pass_count = int(begin), int(count)
        seg = C.c_uint64()
        _lib.check(
            _lib.lib().mp_render_pass_multi(
                self._ctxs, self._hs, len(self.scenes), C.byref(self._sampler), C.byref(st), 1 if gather else 0, self.image.data_ptr(),
                self.image_u8.data_ptr(), C.byref(seg), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream),
            )
        )
        self.segments = seg.value
        nxt = total if count == 0 else begin + count
        return (nxt, self.image, self.image_u8) if gather else nxt


def render_tile(scene: Scene, sampler: CameraSampler, settings: RenderSettings, tile: ScreenBlock):
    """Worker::render_tile (worker.rs:32-49), synchronous: returns (f32 means [h,w,4], u8 [h,w,4])."""
    bvh = scene.object
    if bvh.ctx is None:
        raise _lib.MinipathError(1, "scene is host-only: build it with a Context")
    w, h = tile.width(), tile.height()
    f = np.zeros((max(h, 0), max(w, 0), 4), np.float32)
    u8 = np.zeros((max(h, 0), max(w, 0), 4), np.uint8)
    s = sampler.as_struct()
    st = settings.as_struct()
    _lib.check(
        _lib.lib().mp_render_tile(bvh.ctx.handle, bvh.handle, C.byref(s), C.byref(st), tile.as_struct(), f.ctypes.data, u8.ctypes.data)
    )
    return f, u8


class FrameRenderer:
    """Throughput path: renders a list of tiles into a tile-major HBM buffer in ONE launch on the current torch
    stream (mp_render_tiles_device) and scatters them into an image (mp_untile).  Used by bench.py and by the
    multi-GPU driver; torch only provides the device memory and the stream."""

    def __init__(self, scene: Scene, camera: Camera, settings: RenderSettings, tiles: Optional[Sequence[ScreenBlock]] = None,
                 tile_buf=None):
        import torch

        bvh = scene.object
        if bvh.ctx is None:
            raise _lib.MinipathError(1, "scene is host-only: build it with a Context")
        self.scene, self.settings = scene, settings
        self.ctx: Context = bvh.ctx
        self.device = torch.device("cuda", self.ctx.device_id)
        w, h = settings.resolution
        self.tiles: List[ScreenBlock] = list(tiles) if tiles is not None else tile_ordering(ScreenBlock(0, 0, w, h), settings.tile_size)
        self._tiles_c = (_lib.Block * max(len(self.tiles), 1))(*[t.as_struct() for t in self.tiles])
        self._sampler = camera.build_sampler(settings.resolution).as_struct()
        self._st = settings.as_struct()
        ts = settings.tile_size
        if tile_buf is None:
            tile_buf = torch.zeros((max(len(self.tiles), 1), ts, ts, 4), dtype=torch.float32, device=self.device)
        assert tile_buf.is_contiguous() and tile_buf.shape[0] >= len(self.tiles) and tuple(tile_buf.shape[1:]) == (ts, ts, 4)
        self.tile_buf = tile_buf  # may be a caller-owned shard (multi-GPU gather source)
        self._img = self._img8 = None
        self.samples_per_frame = sum(t.area() for t in self.tiles) * settings.sample_count
        self.rays_per_frame = self.samples_per_frame  # reference semantics: one Object::intersect per sample
        self.segments = torch.zeros(1, dtype=torch.int64, device=self.device)  # ray segments of the last launch
        # profile-guided hand-out order (mp_launch_extras): shader-clock cycles per tile accumulate here; rebalance() turns
        # them into "expensive tiles first" for the following launches (tile i keeps rendering into slot i)
        self.tile_cost = torch.zeros(max(len(self.tiles), 1), dtype=torch.int64, device=self.device)
        self._order_c = None
        self._extras = _lib.LaunchExtras(self.segments.data_ptr(), self.tile_cost.data_ptr(), None)

    def _stream(self):
        import torch

        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def render(self):
        """One pass of the hot path over this renderer's tiles; asynchronous on the current stream."""
        _lib.check(
            _lib.lib().mp_render_tiles_device_ex(
                self.ctx.handle, self.scene.object.handle, C.byref(self._sampler), C.byref(self._st), self._tiles_c,
                len(self.tiles), self.tile_buf.data_ptr(), C.byref(self._extras), self._stream(),
            )
        )
        return self.tile_buf

    def rebalance(self) -> List[int]:
        """Order the following launches' tile hand-out by the measured cost of the launches so far (most expensive first), and
        reset the cost counters.  Synchronises the device (reads `tile_cost`).  The image does not depend on the order; the
        tail of a launch does: waves that finish early find only cheap tiles left."""
        cost = self.tile_cost[: len(self.tiles)].cpu().numpy()
        order = sorted(range(len(self.tiles)), key=lambda i: (-int(cost[i]), i))
        self._order_c = (C.c_uint32 * max(len(order), 1))(*order)
        self._extras.tile_order = C.cast(self._order_c, C.POINTER(C.c_uint32)) if order else None
        self.tile_cost.zero_()
        return order

    def render_pass(self, begin: int, count: int = 0) -> int:
        """Progressive accumulation (MP_FLAG_ACCUMULATE): draws samples [begin, begin+count) of settings.sample_count
        (count 0 = through the last) on top of the running per-pixel sums `tile_buf` holds from the earlier passes, and returns
        the next sample index.  The pass that reaches sample_count leaves the means of worker.rs:44 in `tile_buf`; any split
        of the samples over passes gives the bit-identical frame of one render().  `tile_buf` + the returned index is the
        checkpoint (io.save_checkpoint / io.load_checkpoint)."""
        total = int(self.settings.sample_count)
        if not (0 <= begin < total) or count < 0 or begin + count > total:
            raise ValueError("pass outside [0, sample_count)")
        st = _lib.SettingsStruct.from_buffer_copy(self._st)
        st.flags |= _lib.MP_FLAG_ACCUMULATE
        st.pass_begin, st.pass_count = int(begin), int(count)
        _lib.check(
            _lib.lib().mp_render_tiles_device_ex(
                self.ctx.handle, self.scene.object.handle, C.byref(self._sampler), C.byref(st), self._tiles_c,
                len(self.tiles), self.tile_buf.data_ptr(), C.byref(self._extras), self._stream(),
            )
        )
        return total if count == 0 else begin + count

    def untile(self, tile_buf=None, tiles: Optional[Sequence[ScreenBlock]] = None, want_u8: bool = True, reuse: bool = False,
               preview_samples: Optional[int] = None):
        """Tile-major -> image-major f32 (+ u8 via color_to_image) on the device.  Empty blocks in `tiles` (padding of
        equal-size multi-GPU shards) are skipped by the kernel.  reuse=True writes into this renderer's cached frame
        buffers instead of allocating new ones (every pixel a tile covers is overwritten).  preview_samples=k: the buffer is
        the running state of an unfinished render_pass() sequence after k samples; the image is its preview
        (mp_untile_preview: sums / k), the buffer is left as it is."""
        import torch

        w, h = self.settings.resolution
        buf = self.tile_buf if tile_buf is None else tile_buf
        tl = self.tiles if tiles is None else tiles
        if tiles is None:
            tiles_c = self._tiles_c
        elif isinstance(tiles, tuple) and len(tiles) == 2 and not isinstance(tiles[0], ScreenBlock):
            tl, tiles_c = tiles  # (list, prebuilt ctypes array)
        else:
            tl = list(tiles)
            tiles_c = (_lib.Block * max(len(tl), 1))(*[t.as_struct() for t in tl])
        if reuse and self._img is not None:
            img, img8 = self._img, (self._img8 if want_u8 else None)
        else:
            img = torch.zeros((h, w, 4), dtype=torch.float32, device=self.device)
            img8 = torch.zeros((h, w, 4), dtype=torch.uint8, device=self.device) if want_u8 else None
            if reuse:
                self._img, self._img8 = img, img8
        if preview_samples is not None:
            _lib.check(
                _lib.lib().mp_untile_preview(
                    self.ctx.handle, C.byref(self._st), tiles_c, len(tl), buf.data_ptr(), int(preview_samples), img.data_ptr(),
                    img8.data_ptr() if img8 is not None else None, self._stream(),
                )
            )
            return img, img8
        _lib.check(
            _lib.lib().mp_untile(
                self.ctx.handle, C.byref(self._st), tiles_c, len(tl), buf.data_ptr(), img.data_ptr(),
                img8.data_ptr() if img8 is not None else None, self._stream(),
            )
        )
        return img, img8
