"""ctypes binding of the CPU oracle (oracle/libminipath_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (minipath_amd/) must never import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libminipath_oracle.so")

NO_PRIM = 0xFFFFFFFF


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc, seconds)."""
    src = os.path.join(_HERE, "minipath_oracle.c")
    hdr = os.path.join(_HERE, "minipath_oracle.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr)
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return _SO


class Rng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4)]


class Ray(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("d", C.c_float * 3), ("inv", C.c_float * 3)]


class Sampler(C.Structure):
    _fields_ = [
        ("center", C.c_float * 3),
        ("up", C.c_float * 3),
        ("right", C.c_float * 3),
        ("film_origin_offset", C.c_float * 3),
        ("pixel_scale", C.c_float),
        ("lens_radius", C.c_float),
        ("lens_weight", C.c_float),
    ]

    def as_array(self) -> np.ndarray:
        return np.frombuffer(bytes(self), dtype=np.float32).copy()


class Camera(C.Structure):
    _fields_ = [
        ("q", C.c_float * 4),
        ("t", C.c_float * 3),
        ("focus_distance", C.c_float),
        ("sensor_is_width", C.c_int),
        ("sensor_size", C.c_float),
        ("focal_length", C.c_float),
        ("f_number", C.c_float),
    ]


class Hit(C.Structure):
    _fields_ = [
        ("hit", C.c_int),
        ("prim", C.c_uint64),
        ("t", C.c_float),
        ("u", C.c_float),
        ("v", C.c_float),
        ("gn", C.c_float * 3),
        ("point", C.c_float * 3),
        ("normal", C.c_float * 3),
        ("tex", C.c_float * 3),
        ("material", C.c_uint64),
        ("instance", C.c_uint32),
    ]


class Counters(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("inner_visited", C.c_uint64),
        ("packets_tested", C.c_uint64),
        ("stack_pops", C.c_uint64),
        ("max_stack", C.c_uint64),
    ]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    f32p = C.POINTER(C.c_float)
    u32p = C.POINTER(C.c_uint32)
    L.mpo_rng_seed.argtypes = [C.POINTER(Rng), C.c_uint64]
    L.mpo_rng_next_u64.argtypes = [C.POINTER(Rng)]
    L.mpo_rng_next_u64.restype = C.c_uint64
    L.mpo_rng_next_u32.argtypes = [C.POINTER(Rng)]
    L.mpo_rng_next_u32.restype = C.c_uint32
    L.mpo_rng_range_pm_half.argtypes = [C.POINTER(Rng)]
    L.mpo_rng_range_pm_half.restype = C.c_float
    L.mpo_rng_unit_disc.argtypes = [C.POINTER(Rng), f32p]
    L.mpo_sample_key.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.mpo_sample_key.restype = C.c_uint64
    L.mpo_ray_new.argtypes = [f32p, f32p, C.POINTER(Ray)]
    L.mpo_ray_point_at.argtypes = [C.POINTER(Ray), C.c_float, f32p]
    L.mpo_aabb8_intersect.argtypes = [f32p, f32p, C.POINTER(Ray), C.c_float, f32p, f32p]
    L.mpo_tri8_intersect.argtypes = [f32p, f32p, f32p, C.POINTER(Ray), f32p, f32p, f32p]
    L.mpo_tri8_intersect.restype = C.c_uint
    L.mpo_unit_interval_compress.argtypes = [C.c_float, C.c_int, C.c_int]
    L.mpo_unit_interval_compress.restype = C.c_uint16
    L.mpo_unit_interval_decompress.argtypes = [C.c_uint16]
    L.mpo_unit_interval_decompress.restype = C.c_float
    L.mpo_bit_iter.argtypes = [C.c_uint64, C.POINTER(C.c_int)]
    L.mpo_bit_iter.restype = C.c_int
    L.mpo_link_new_leaf.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_int)]
    L.mpo_link_new_leaf.restype = C.c_uint32
    L.mpo_link_new_inner.argtypes = [C.c_uint32, C.POINTER(C.c_int)]
    L.mpo_link_new_inner.restype = C.c_uint32
    L.mpo_link_decode.argtypes = [C.c_uint32, u32p, u32p]
    L.mpo_link_decode.restype = C.c_int
    L.mpo_camera_default.argtypes = [C.POINTER(Camera)]
    L.mpo_camera_look_at.argtypes = [C.POINTER(Camera), f32p, f32p, f32p]
    L.mpo_camera_look_direction.argtypes = [C.POINTER(Camera), f32p, f32p, f32p]
    L.mpo_camera_translate.argtypes = [C.POINTER(Camera), f32p]
    L.mpo_camera_basis.argtypes = [C.POINTER(Camera), f32p, f32p, f32p, f32p]
    L.mpo_camera_build_sampler.argtypes = [C.POINTER(Camera), C.c_uint32, C.c_uint32, C.POINTER(Sampler)]
    L.mpo_sample_ray.argtypes = [C.POINTER(Sampler), C.c_uint32, C.c_uint32, C.POINTER(Rng), C.POINTER(Ray)]
    L.mpo_divide_range.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u32p, C.c_size_t]
    L.mpo_divide_range.restype = C.c_size_t
    L.mpo_tile_ordering.argtypes = [C.c_uint32] * 5 + [C.c_uint64, u32p, C.c_size_t]
    L.mpo_tile_ordering.restype = C.c_size_t
    L.mpo_internal_points.argtypes = [C.c_uint32] * 4 + [u32p, C.c_size_t]
    L.mpo_internal_points.restype = C.c_size_t
    L.mpo_bvh_from_obj.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.mpo_bvh_from_obj.restype = C.c_void_p
    L.mpo_bvh_build.argtypes = [f32p, f32p, f32p, C.c_uint32, u32p, C.c_uint32, C.c_char_p, C.c_size_t]
    L.mpo_bvh_build.restype = C.c_void_p
    L.mpo_bvh_build_mat.argtypes = [f32p, f32p, f32p, C.c_uint32, u32p, u32p, C.c_uint32, C.c_char_p, C.c_size_t]
    L.mpo_bvh_build_mat.restype = C.c_void_p
    L.mpo_bvh_from_arrays.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, u32p, f32p, f32p, C.c_uint32,
                                      C.c_uint32, f32p, f32p, C.c_char_p, C.c_size_t]
    L.mpo_bvh_from_arrays.restype = C.c_void_p
    L.mpo_bvh_set_materials.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float]
    L.mpo_bvh_set_materials.restype = C.c_int
    L.mpo_bvh_tri_material.argtypes = [C.c_void_p]
    L.mpo_bvh_tri_material.restype = C.c_void_p
    L.mpo_bvh_material_count.argtypes = [C.c_void_p]
    L.mpo_bvh_material_count.restype = C.c_uint32
    L.mpo_set_chunked_sum.argtypes = [C.c_int]
    L.mpo_bvh_set_instances.argtypes = [C.c_void_p, f32p, C.c_uint32]
    L.mpo_bvh_set_instances.restype = C.c_int
    L.mpo_bvh_set_group.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), f32p, f32p, C.c_uint32]
    L.mpo_bvh_set_group.restype = C.c_int
    L.mpo_bvh_set_group_rotations.argtypes = [C.c_void_p, f32p]
    L.mpo_bvh_set_group_rotations.restype = C.c_int
    L.mpo_trace_rays_inst.argtypes = [C.c_void_p] + [f32p] * 6 + [C.c_uint64, f32p, u32p, f32p, f32p, u32p]
    L.mpo_seed_mix.argtypes = [C.c_uint64]
    L.mpo_seed_mix.restype = C.c_uint64
    L.mpo_bvh_free.argtypes = [C.c_void_p]
    for name in ("root", "inner_count", "packet_count", "vertex_count", "depth"):
        fn = getattr(L, f"mpo_bvh_{name}")
        fn.argtypes = [C.c_void_p]
        fn.restype = C.c_uint32
    L.mpo_bvh_bbox.argtypes = [C.c_void_p, f32p, f32p]
    for name in ("inner_nodes", "packets", "tri_shading", "vertex_normals", "vertex_tex"):
        fn = getattr(L, f"mpo_bvh_{name}")
        fn.argtypes = [C.c_void_p]
        fn.restype = C.c_void_p
    L.mpo_bvh_intersect.argtypes = [C.c_void_p, C.POINTER(Ray), C.POINTER(Hit), C.POINTER(Counters)]
    L.mpo_bvh_intersect_ops.argtypes = [C.c_void_p, C.POINTER(Ray), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.c_size_t]
    L.mpo_bvh_intersect_ops.restype = C.c_size_t
    L.mpo_trace_rays.argtypes = [C.c_void_p] + [f32p] * 6 + [C.c_uint64, f32p, u32p, f32p, f32p, C.POINTER(Counters)]
    L.mpo_render_sample.argtypes = [
        C.c_void_p, C.POINTER(Sampler), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
        f32p, C.POINTER(Counters),
    ]
    L.mpo_render_tile.argtypes = [
        C.c_void_p, C.POINTER(Sampler), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
        C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f32p, C.POINTER(C.c_uint8), C.POINTER(Counters),
    ]
    L.mpo_color_to_image.argtypes = [f32p, C.POINTER(C.c_uint8)]
    L.mpo_render_image_mt.argtypes = [
        C.c_void_p, C.POINTER(Sampler), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int,
        C.c_size_t, C.c_size_t, f32p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(Counters),
    ]
    L.mpo_render_image_mt.restype = C.c_double
    L.mpo_sphere_intersect.argtypes = [f32p, C.c_float, C.POINTER(Ray), C.POINTER(Hit)]
    L.mpo_render_tile_sphere.argtypes = [
        f32p, C.c_float, C.POINTER(Sampler), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
        f32p, C.POINTER(C.c_uint8),
    ]
    L.mpo_render_tile_paths.argtypes = [
        C.c_void_p, C.POINTER(Sampler), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32,
        C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f32p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64),
    ]
    L.mpo_render_image_paths_mt.argtypes = [
        C.c_void_p, C.POINTER(Sampler), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int,
        C.c_size_t, C.c_size_t, f32p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64),
    ]
    L.mpo_render_image_paths_mt.restype = C.c_double
    _lib = L
    return L


def material_records(table) -> np.ndarray:
    """mpo_material records (48 B: albedo rgb, emission rgb, albedo2 rgb, texture u32, tex_scale f32, pad) from a table of
    (albedo, emission) pairs -- scalars = grey, triples = rgb -- or dicts with "albedo", "emission", "albedo2", "checker"."""
    def rgb(x):
        a = np.asarray(x, np.float32).reshape(-1)
        return np.repeat(a, 3) if a.size == 1 else a.reshape(3)

    rec = np.zeros((len(table), 12), np.float32)
    for i, e in enumerate(table):
        if isinstance(e, dict):
            rec[i, 0:3] = rgb(e.get("albedo", 0.75))
            rec[i, 3:6] = rgb(e.get("emission", 0.0))
            rec[i, 6:9] = rgb(e.get("albedo2", e.get("albedo", 0.75)))
            if e.get("checker") is not None:
                rec[i, 9:10].view(np.uint32)[0] = 1
                rec[i, 10] = np.float32(e["checker"])
        else:
            rec[i, 0:3] = rgb(e[0])
            rec[i, 3:6] = rgb(e[1])
            rec[i, 6:9] = rec[i, 0:3]
    return rec


def _f32p(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32p(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def vec3(x, y, z):
    return (C.c_float * 3)(x, y, z)


def ray_new(o, d) -> Ray:
    r = Ray()
    lib().mpo_ray_new(vec3(*o), vec3(*d), C.byref(r))
    return r


def point_at(r: Ray, t: float):
    out = (C.c_float * 3)()
    lib().mpo_ray_point_at(C.byref(r), C.c_float(t), out)
    return [out[0], out[1], out[2]]


def aabb8_intersect(bmin: np.ndarray, bmax: np.ndarray, ray: Ray, max_t: float):
    """bmin,bmax: float32 [3][8]."""
    t1 = np.zeros(8, np.float32)
    t2 = np.zeros(8, np.float32)
    lib().mpo_aabb8_intersect(_f32p(bmin), _f32p(bmax), C.byref(ray), C.c_float(max_t), _f32p(t1), _f32p(t2))
    return t1, t2


def tri8_intersect(v0, v1, v2, ray: Ray):
    t = np.zeros(8, np.float32)
    u = np.zeros(8, np.float32)
    v = np.zeros(8, np.float32)
    m = lib().mpo_tri8_intersect(_f32p(v0), _f32p(v1), _f32p(v2), C.byref(ray), _f32p(t), _f32p(u), _f32p(v))
    return m, t, u, v


def teapot_camera() -> Camera:
    """benches/render_teapot.rs:12-19."""
    c = Camera()
    L = lib()
    L.mpo_camera_default(C.byref(c))
    L.mpo_camera_look_at(C.byref(c), vec3(0.0, 2.0, 10.0), vec3(0.0, 1.5, 0.0), vec3(0.0, 1.0, 0.0))
    c.f_number = 4.8
    c.focus_distance = 10.0
    return c


def build_sampler(cam: Camera, w: int, h: int) -> Sampler:
    s = Sampler()
    lib().mpo_camera_build_sampler(C.byref(cam), w, h, C.byref(s))
    return s


def sampler_from_array(a) -> Sampler:
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.size == 15
    s = Sampler()
    C.memmove(C.byref(s), a.ctypes.data, 60)
    return s


class Bvh:
    """Owns an mpo_bvh*."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle BVH build failed")
        self.h = C.c_void_p(handle)

    @classmethod
    def from_obj(cls, path: str) -> "Bvh":
        err = C.create_string_buffer(512)
        h = lib().mpo_bvh_from_obj(path.encode(), err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return cls(h)

    @classmethod
    def build(cls, pos, nrm, tex, tri, tri_material=None) -> "Bvh":
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        nv = pos.shape[0]
        nrm = None if nrm is None else np.ascontiguousarray(nrm, np.float32).reshape(nv, 3)
        tex = None if tex is None else np.ascontiguousarray(tex, np.float32).reshape(nv, 3)
        tri = np.ascontiguousarray(tri, np.uint32).reshape(-1, 3)
        mat = None if tri_material is None else np.ascontiguousarray(tri_material, np.uint32).reshape(tri.shape[0])
        err = C.create_string_buffer(512)
        h = lib().mpo_bvh_build_mat(
            _f32p(pos), None if nrm is None else _f32p(nrm), None if tex is None else _f32p(tex), nv,
            _u32p(tri), None if mat is None else _u32p(mat), tri.shape[0], err, 512,
        )
        if not h:
            raise RuntimeError(err.value.decode())
        return cls(h)

    @classmethod
    def from_arrays(cls, inner, packets, shading, vnormal, vtex, root, bmin, bmax, material=None) -> "Bvh":
        """A TriangleBvh given as its reference-layout arrays (what mp_scene_export emits / mp_scene_from_arrays takes):
        inner (n,128) u8, packets (n,144) u8, shading (n*8,4) u32, vertex normals / tex (nv,3) f32, material (n*8) u32."""
        inner = np.ascontiguousarray(inner, np.uint8).reshape(-1, 128)
        packets = np.ascontiguousarray(packets, np.uint8).reshape(-1, 144)
        shading = np.ascontiguousarray(shading, np.uint32).reshape(-1, 4)
        vn = np.ascontiguousarray(vnormal, np.float32).reshape(-1, 3)
        vt = np.ascontiguousarray(vtex, np.float32).reshape(-1, 3)
        mat = None if material is None else np.ascontiguousarray(material, np.uint32).reshape(-1)
        bmin = np.ascontiguousarray(bmin, np.float32)
        bmax = np.ascontiguousarray(bmax, np.float32)
        err = C.create_string_buffer(512)
        h = lib().mpo_bvh_from_arrays(inner.ctypes.data, inner.shape[0], packets.ctypes.data, packets.shape[0], shading.ctypes.data,
                                      None if mat is None else _u32p(mat), _f32p(vn), _f32p(vt), vn.shape[0], int(root),
                                      _f32p(bmin), _f32p(bmax), err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return cls(h)

    def set_materials(self, table, sky: float = 1.0) -> None:
        """Material table + sky radiance of the build-defined path extension.  Entries: (albedo, emission) with scalars (grey) or
        rgb triples, or a dict {"albedo", "emission", "albedo2", "checker": cells per unit of texture coordinate}."""
        t = material_records(table)
        if not lib().mpo_bvh_set_materials(self.h, t.ctypes.data_as(C.c_void_p), t.shape[0], C.c_float(sky)):
            raise RuntimeError("material id of a triangle outside the table")

    def set_instances(self, translations) -> None:
        """BUILD-DEFINED Object: translated instances of this BVH ([] restores the plain TriangleBvh)."""
        t = np.ascontiguousarray(translations, np.float32).reshape(-1, 3)
        if not lib().mpo_bvh_set_instances(self.h, _f32p(t) if t.shape[0] else None, t.shape[0]):
            raise RuntimeError("set_instances failed")

    def set_group(self, objects, translations, rotations=None) -> None:
        """BUILD-DEFINED Object: members {objects[k], translation k}; a member is a Bvh or a sphere given as (center, radius).
        This BVH is the container (its materials and sky apply).  Keeps references to the members."""
        t = np.ascontiguousarray(translations, np.float32).reshape(-1, 3)
        objects = list(objects)
        sph = np.full((len(objects), 4), -1.0, np.float32)
        for k, o in enumerate(objects):
            if not isinstance(o, Bvh):
                sph[k, :3], sph[k, 3] = o[0], o[1]
        arr = (C.c_void_p * len(objects))(*[o.h if isinstance(o, Bvh) else None for o in objects])
        has_sphere = bool((sph[:, 3] >= 0).any())
        if len(objects) != t.shape[0] or not lib().mpo_bvh_set_group(self.h, arr, _f32p(sph) if has_sphere else None, _f32p(t), t.shape[0]):
            raise RuntimeError("set_group failed")
        self._members = objects
        if rotations is not None:
            q = np.ascontiguousarray(rotations, np.float32).reshape(-1, 4)
            if q.shape[0] != t.shape[0] or not lib().mpo_bvh_set_group_rotations(self.h, _f32p(q)):
                raise RuntimeError("set_group rotations failed")

    def trace_inst(self, o: np.ndarray, d: np.ndarray):
        """trace() plus the instance index of every hit."""
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
        t = np.zeros(n, np.float32); prim = np.zeros(n, np.uint32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        inst = np.zeros(n, np.uint32)
        lib().mpo_trace_rays_inst(self.h, *[_f32p(c) for c in cols], n, _f32p(t), _u32p(prim), _f32p(u), _f32p(v), _u32p(inst))
        return t, prim, u, v, inst

    def tri_material(self) -> np.ndarray:
        return self._view(lib().mpo_bvh_tri_material(self.h), self.n_packets * 8 * 4, np.uint32).copy()

    @property
    def material_count(self) -> int:
        return lib().mpo_bvh_material_count(self.h)

    def __del__(self):
        try:
            if self.h:
                lib().mpo_bvh_free(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def root(self) -> int:
        return lib().mpo_bvh_root(self.h)

    @property
    def n_inner(self) -> int:
        return lib().mpo_bvh_inner_count(self.h)

    @property
    def n_packets(self) -> int:
        return lib().mpo_bvh_packet_count(self.h)

    @property
    def n_vertices(self) -> int:
        return lib().mpo_bvh_vertex_count(self.h)

    @property
    def depth(self) -> int:
        return lib().mpo_bvh_depth(self.h)

    def bbox(self):
        a = np.zeros(3, np.float32)
        b = np.zeros(3, np.float32)
        lib().mpo_bvh_bbox(self.h, _f32p(a), _f32p(b))
        return a, b

    def _view(self, ptr, nbytes, dtype):
        if nbytes == 0:
            return np.zeros(0, dtype)
        buf = (C.c_uint8 * nbytes).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).copy()

    def inner_nodes_bytes(self) -> np.ndarray:
        return self._view(lib().mpo_bvh_inner_nodes(self.h), self.n_inner * 128, np.uint8)

    def packets_bytes(self) -> np.ndarray:
        return self._view(lib().mpo_bvh_packets(self.h), self.n_packets * 144, np.uint8)

    def tri_shading(self) -> np.ndarray:
        return self._view(lib().mpo_bvh_tri_shading(self.h), self.n_packets * 8 * 16, np.uint32).reshape(-1, 4)

    def vertex_normals(self) -> np.ndarray:
        return self._view(lib().mpo_bvh_vertex_normals(self.h), self.n_vertices * 12, np.float32).reshape(-1, 3)

    def vertex_tex(self) -> np.ndarray:
        return self._view(lib().mpo_bvh_vertex_tex(self.h), self.n_vertices * 12, np.float32).reshape(-1, 3)

    def intersect(self, ray: Ray, counters: Counters | None = None) -> Hit:
        h = Hit()
        lib().mpo_bvh_intersect(self.h, C.byref(ray), C.byref(h), C.byref(counters) if counters is not None else None)
        return h

    def trace(self, o: np.ndarray, d: np.ndarray, counters: Counters | None = None):
        """o, d: float32 [n,3] (d need not be unit). Returns t, prim(u32, NO_PRIM=miss), u, v."""
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
        t = np.zeros(n, np.float32)
        prim = np.zeros(n, np.uint32)
        u = np.zeros(n, np.float32)
        v = np.zeros(n, np.float32)
        lib().mpo_trace_rays(
            self.h, *[_f32p(c) for c in cols], n, _f32p(t), _u32p(prim), _f32p(u), _f32p(v),
            C.byref(counters) if counters is not None else None,
        )
        return t, prim, u, v

    def render_tile(self, sampler: Sampler, width, height, spp, seed, x0, y0, x1, y1, counters=None):
        tw, th = x1 - x0, y1 - y0
        f = np.zeros((th, tw, 4), np.float32)
        u8 = np.zeros((th, tw, 4), np.uint8)
        lib().mpo_render_tile(
            self.h, C.byref(sampler), width, height, spp, C.c_uint64(seed), x0, y0, x1, y1, _f32p(f),
            u8.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(counters) if counters is not None else None,
        )
        return f, u8

    def render_tile_paths(self, sampler: Sampler, width, height, spp, seed, max_depth, x0, y0, x1, y1):
        """Build-defined path extension.  Returns f32 [h,w,4], u8, ray segments traced."""
        tw, th = x1 - x0, y1 - y0
        f = np.zeros((th, tw, 4), np.float32)
        u8 = np.zeros((th, tw, 4), np.uint8)
        seg = C.c_uint64(0)
        lib().mpo_render_tile_paths(
            self.h, C.byref(sampler), width, height, spp, C.c_uint64(seed), max_depth, x0, y0, x1, y1, _f32p(f),
            u8.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(seg),
        )
        return f, u8, seg.value

    def render_image_paths_mt(self, sampler: Sampler, width, height, spp, seed, max_depth, tile=64, nthreads=1, max_tiles=0,
                              tile_stride=1):
        f = np.zeros((height, width, 4), np.float32)
        u8 = np.zeros((height, width, 4), np.uint8)
        seg = C.c_uint64(0)
        secs = lib().mpo_render_image_paths_mt(
            self.h, C.byref(sampler), width, height, spp, C.c_uint64(seed), max_depth, tile, nthreads, max_tiles, tile_stride,
            _f32p(f), u8.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(seg),
        )
        return f, u8, secs, seg.value

    def render_image_mt(self, sampler: Sampler, width, height, spp, seed, tile=64, nthreads=1, max_tiles=0,
                        tile_stride=1, want_counters=False):
        f = np.zeros((height, width, 4), np.float32)
        u8 = np.zeros((height, width, 4), np.uint8)
        rays = C.c_uint64(0)
        cnt = Counters()
        secs = lib().mpo_render_image_mt(
            self.h, C.byref(sampler), width, height, spp, C.c_uint64(seed), tile, nthreads, max_tiles, tile_stride,
            _f32p(f), u8.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(rays), C.byref(cnt) if want_counters else None,
        )
        return f, u8, secs, rays.value, cnt


def sphere_intersect(center, radius, ray: Ray) -> Hit:
    h = Hit()
    lib().mpo_sphere_intersect(vec3(*center), C.c_float(radius), C.byref(ray), C.byref(h))
    return h


def render_tile_sphere(center, radius, sampler: Sampler, width, spp, seed, x0, y0, x1, y1):
    f = np.zeros((y1 - y0, x1 - x0, 4), np.float32)
    u8 = np.zeros((y1 - y0, x1 - x0, 4), np.uint8)
    lib().mpo_render_tile_sphere(vec3(*center), C.c_float(radius), C.byref(sampler), width, spp, C.c_uint64(seed), x0, y0, x1, y1,
                                 _f32p(f), u8.ctypes.data_as(C.POINTER(C.c_uint8)))
    return f, u8


def sample_ray(sampler: Sampler, x: int, y: int, key: int) -> Ray:
    rng = Rng()
    lib().mpo_rng_seed(C.byref(rng), C.c_uint64(key))
    r = Ray()
    lib().mpo_sample_ray(C.byref(sampler), x, y, C.byref(rng), C.byref(r))
    return r


def tile_ordering(minx, miny, maxx, maxy, tile, shuffle_seed=0) -> np.ndarray:
    n = lib().mpo_tile_ordering(minx, miny, maxx, maxy, tile, C.c_uint64(shuffle_seed), None, 0)
    out = np.zeros((n, 4), np.uint32)
    if n:
        lib().mpo_tile_ordering(minx, miny, maxx, maxy, tile, C.c_uint64(shuffle_seed), _u32p(out), n)
    return out


def internal_points(minx, miny, maxx, maxy) -> np.ndarray:
    n = lib().mpo_internal_points(minx, miny, maxx, maxy, None, 0)
    out = np.zeros((n, 2), np.uint32)
    if n:
        lib().mpo_internal_points(minx, miny, maxx, maxy, _u32p(out), n)
    return out
