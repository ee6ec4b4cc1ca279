"""ctypes loader for libminipath_hip.so (the C ABI of include/minipath_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (or ``make -C minipath_amd/csrc``).  There is no
CPU fallback: if the shared object is missing, importing any compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MINIPATH_HIP_SO: experiment builds of the same library (kernel variants under measurement); default = the in-tree build
SO_PATH = os.environ.get("MINIPATH_HIP_SO") or os.path.join(_HERE, "csrc", "libminipath_hip.so")

MP_OK = 0
MP_NO_PRIM = 0xFFFFFFFF
MP_LINK_NULL = 0xFFFFFFF8
MP_FLAG_SHUFFLE_TILES = 1
MP_FLAG_TRAVERSAL_GROUPS = 2
MP_FLAG_PATHS = 4
MP_FLAG_ACCUMULATE = 8
MP_FLAG_WAVEFRONT = 16
MP_FLAG_CHUNKED_SUM = 32
MP_FLAG_IMAGE_U8_ONLY = 64


MAX_MATERIALS = 65536  # MP_MAX_MATERIALS (include/minipath_hip.h)


class MinipathError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"minipath_hip error {code}: {message}")
        self.code = code
        self.message = message


class Block(C.Structure):
    """geometry/mod.rs:15 ScreenBlock = [min, max)."""

    _fields_ = [("min_x", C.c_uint32), ("min_y", C.c_uint32), ("max_x", C.c_uint32), ("max_y", C.c_uint32)]

    def as_tuple(self):
        return (self.min_x, self.min_y, self.max_x, self.max_y)


class CameraStruct(C.Structure):
    _fields_ = [
        ("q", C.c_float * 4),
        ("t", C.c_float * 3),
        ("focus_distance", C.c_float),
        ("sensor_is_width", C.c_int32),
        ("sensor_size", C.c_float),
        ("focal_length", C.c_float),
        ("f_number", C.c_float),
    ]


class SamplerStruct(C.Structure):
    """camera.rs:26-39 CameraSampler."""

    _fields_ = [
        ("center", C.c_float * 3),
        ("up", C.c_float * 3),
        ("right", C.c_float * 3),
        ("film_origin_offset", C.c_float * 3),
        ("pixel_scale", C.c_float),
        ("lens_radius", C.c_float),
        ("lens_weight", C.c_float),
    ]


class SettingsStruct(C.Structure):
    _fields_ = [
        ("tile_size", C.c_uint32),
        ("sample_count", C.c_uint32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("seed", C.c_uint64),
        ("flags", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("pass_begin", C.c_uint32),
        ("pass_count", C.c_uint32),
    ]


class Progress(C.Structure):
    _fields_ = [("finished", C.c_size_t), ("total", C.c_size_t)]


class SceneInfo(C.Structure):
    _fields_ = [
        ("root_link", C.c_uint32),
        ("inner_count", C.c_uint32),
        ("packet_count", C.c_uint32),
        ("vertex_count", C.c_uint32),
        ("triangle_count", C.c_uint32),
        ("depth", C.c_uint32),
        ("stack_bound", C.c_uint32),
        ("bbox_min", C.c_float * 3),
        ("bbox_max", C.c_float * 3),
        ("device_bytes", C.c_uint64),
        ("material_count", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class BvhDesc(C.Structure):
    """mp_bvh_desc: a TriangleBvh as arrays in the reference's own layout (triangle_bvh/mod.rs:20-53)."""

    _fields_ = [
        ("inner_nodes", C.c_void_p),
        ("packets", C.c_void_p),
        ("tri_shading", C.c_void_p),
        ("tri_material", C.c_void_p),
        ("vertex_normals", C.c_void_p),
        ("vertex_tex", C.c_void_p),
        ("inner_count", C.c_uint32),
        ("packet_count", C.c_uint32),
        ("vertex_count", C.c_uint32),
        ("root_link", C.c_uint32),
        ("bbox_min", C.c_float * 3),
        ("bbox_max", C.c_float * 3),
    ]


class Material(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("emission", C.c_float * 3), ("albedo2", C.c_float * 3), ("texture", C.c_uint32),
                ("texture_scale", C.c_float), ("reserved", C.c_uint32)]

    @classmethod
    def make(cls, entry) -> "Material":
        """(albedo, emission) with scalars (grey) or rgb triples, or a dict {"albedo", "emission", "albedo2", "checker": cells per
        unit of texture coordinate}."""
        def rgb(x):
            try:
                v = [float(c) for c in x]
            except TypeError:
                v = [float(x)] * 3
            if len(v) == 1:
                v = v * 3
            if len(v) != 3:
                raise ValueError("a colour is a scalar or an (r, g, b) triple")
            return (C.c_float * 3)(*v)

        if isinstance(entry, dict):
            a = entry.get("albedo", 0.75)
            m = cls(rgb(a), rgb(entry.get("emission", 0.0)), rgb(entry.get("albedo2", a)), 0, 0.0, 0)
            if entry.get("checker") is not None:
                m.texture, m.texture_scale = MP_TEXTURE_CHECKER, float(entry["checker"])
            return m
        a, e = entry
        return cls(rgb(a), rgb(e), rgb(a), 0, 0.0, 0)


MP_TEXTURE_NONE, MP_TEXTURE_CHECKER = 0, 1


class HitsSoA(C.Structure):
    _fields_ = [
        ("d_t", C.c_void_p),
        ("d_prim", C.c_void_p),
        ("d_u", C.c_void_p),
        ("d_v", C.c_void_p),
        ("d_point", C.c_void_p),
        ("d_normal", C.c_void_p),
        ("d_tex", C.c_void_p),
        ("d_material", C.c_void_p),
        ("d_instance", C.c_void_p),
    ]


class LaunchExtras(C.Structure):
    _fields_ = [("d_ray_segments", C.c_void_p), ("d_tile_cost", C.c_void_p), ("tile_order", C.POINTER(C.c_uint32))]


STARTED_CB = C.CFUNCTYPE(None, C.c_void_p, Block)
FINISHED_CB = C.CFUNCTYPE(None, C.c_void_p, Block, Progress)

# name -> (restype, argtypes); every symbol include/minipath_hip.h declares
_f3 = C.POINTER(C.c_float)
SIGNATURES = {
    "mp_last_error": (C.c_char_p, []),
    "mp_version": (C.c_char_p, []),
    "mp_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mp_ctx_destroy": (None, [C.c_void_p]),
    "mp_ctx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "mp_ctx_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mp_camera_default": (C.c_int, [C.POINTER(CameraStruct)]),
    "mp_camera_look_at": (C.c_int, [C.POINTER(CameraStruct), _f3, _f3, _f3]),
    "mp_camera_look_direction": (C.c_int, [C.POINTER(CameraStruct), _f3, _f3, _f3]),
    "mp_camera_translate": (C.c_int, [C.POINTER(CameraStruct), _f3]),
    "mp_camera_basis": (C.c_int, [C.POINTER(CameraStruct), _f3, _f3, _f3, _f3]),
    "mp_camera_build_sampler": (C.c_int, [C.POINTER(CameraStruct), C.c_uint32, C.c_uint32, C.POINTER(SamplerStruct)]),
    "mp_tile_ordering": (C.c_int, [Block, C.c_uint32, C.c_uint64, C.POINTER(Block), C.c_size_t, C.POINTER(C.c_size_t)]),
    "mp_scene_from_obj": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "mp_scene_from_triangles": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)],
    ),
    "mp_scene_from_triangles_mat": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)],
    ),
    "mp_scene_from_arrays": (C.c_int, [C.c_void_p, C.POINTER(BvhDesc), C.POINTER(C.c_void_p)]),
    "mp_scene_set_materials": (C.c_int, [C.c_void_p, C.POINTER(Material), C.c_uint32, C.c_float]),
    "mp_scene_material_name": (C.c_char_p, [C.c_void_p, C.c_uint32]),
    "mp_scene_sphere": (C.c_int, [C.c_void_p, _f3, C.c_float, C.POINTER(C.c_void_p)]),
    "mp_scene_instances": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "mp_scene_group": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "mp_scene_destroy": (None, [C.c_void_p]),
    "mp_scene_info_get": (C.c_int, [C.c_void_p, C.POINTER(SceneInfo)]),
    "mp_scene_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mp_scene_device_tree": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mp_trace_rays": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_void_p] * 6 + [C.c_uint64, C.POINTER(HitsSoA), C.c_void_p]),
    "mp_generate_rays": (
        C.c_int,
        [C.c_void_p, C.POINTER(SamplerStruct), C.POINTER(SettingsStruct), Block, C.c_uint32] + [C.c_void_p] * 6 + [C.c_void_p],
    ),
    "mp_render_tile": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(SamplerStruct), C.POINTER(SettingsStruct), Block, C.c_void_p, C.c_void_p],
    ),
    "mp_render_tiles_device": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(SamplerStruct), C.POINTER(SettingsStruct), C.POINTER(Block), C.c_size_t,
         C.c_void_p, C.c_void_p],
    ),
    "mp_render_tiles_device_counted": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(SamplerStruct), C.POINTER(SettingsStruct), C.POINTER(Block), C.c_size_t,
         C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "mp_render_tiles_device_ex": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(SamplerStruct), C.POINTER(SettingsStruct), C.POINTER(Block), C.c_size_t,
         C.c_void_p, C.POINTER(LaunchExtras), C.c_void_p],
    ),
    "mp_untile": (
        C.c_int,
        [C.c_void_p, C.POINTER(SettingsStruct), C.POINTER(Block), C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "mp_render_begin": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(CameraStruct), C.POINTER(SettingsStruct), STARTED_CB, FINISHED_CB, C.c_void_p,
         C.POINTER(C.c_void_p)],
    ),
    "mp_render_begin_multi": (
        C.c_int,
        [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.POINTER(CameraStruct), C.POINTER(SettingsStruct), STARTED_CB,
         FINISHED_CB, C.c_void_p, C.POINTER(C.c_void_p)],
    ),
    "mp_render_frame_multi": (
        C.c_int,
        [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.POINTER(SamplerStruct), C.POINTER(SettingsStruct), C.c_void_p,
         C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p],
    ),
    "mp_render_pass_multi": (
        C.c_int,
        [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.POINTER(SamplerStruct), C.POINTER(SettingsStruct), C.c_int, C.c_void_p,
         C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p],
    ),
    "mp_untile_preview": (C.c_int, [C.c_void_p, C.POINTER(SettingsStruct), C.POINTER(Block), C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "mp_render_progress": (C.c_int, [C.c_void_p, C.POINTER(Progress)]),
    "mp_render_is_finished": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "mp_render_elapsed_ns": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mp_render_abort": (C.c_int, [C.c_void_p]),
    "mp_render_wait": (C.c_int, [C.c_void_p]),
    "mp_render_image_u8": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mp_render_image_f32": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mp_render_destroy": (None, [C.c_void_p]),
}

_lib = None


def lib() -> C.CDLL:
    """Load the HIP library.  Fails loudly when it has not been built: there is no other compute path."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C minipath_amd/csrc` -- minipath_amd has no CPU fallback"
            )
        try:
            # torch ships its own libamdhip64; the process must hold ONE HIP runtime, so let torch's load first when it
            # is installed (otherwise a later torch.cuda initialisation finds "No HIP GPUs")
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != MP_OK:
        msg = lib().mp_last_error()
        raise MinipathError(rc, msg.decode() if msg else "")
