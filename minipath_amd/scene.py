"""Mirror of src/scene: Scene<O> / TriangleBvh (scene/mod.rs:7-15, scene/triangle_bvh/mod.rs:20-30)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib


class Context:
    """One GPU (include/minipath_hip.h: one context drives one device)."""

    def __init__(self, device_id: int = 0):
        h = C.c_void_p()
        _lib.check(_lib.lib().mp_ctx_create(int(device_id), C.byref(h)))
        self.handle = h
        self.device_id = int(device_id)

    @property
    def cu_count(self) -> int:
        cu = C.c_int()
        _lib.check(_lib.lib().mp_ctx_device(self.handle, None, C.byref(cu)))
        return cu.value

    def set_option(self, key: str, value: int) -> None:
        """mp_ctx_set_option: tuning knobs (results never depend on them)."""
        _lib.check(_lib.lib().mp_ctx_set_option(self.handle, key.encode(), int(value)))

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().mp_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TriangleBvh:
    """scene/triangle_bvh/mod.rs:20-30.  Built on the host (building.rs), resident in HBM when a Context is given."""

    def __init__(self, handle, ctx: Optional[Context]):
        self.handle = handle
        self.ctx = ctx

    @classmethod
    def with_obj(cls, path: str, ctx: Optional[Context] = None) -> "TriangleBvh":
        """TriangleBvh::with_obj (building.rs:28-34).  ctx=None builds a host-only BVH (no GPU needed)."""
        h = C.c_void_p()
        _lib.check(_lib.lib().mp_scene_from_obj(ctx.handle if ctx else None, str(path).encode(), C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def build(cls, positions, normals, tex, triangles, ctx: Optional[Context] = None, tri_material=None) -> "TriangleBvh":
        """TriangleBvh::build (building.rs:83-107) over indexed triangles.  tri_material: optional material id per triangle
        (the reference writes `material: 0`, building.rs:201)."""
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        nv = pos.shape[0]
        nrm = None if normals is None else np.ascontiguousarray(normals, np.float32).reshape(nv, 3)
        tx = None if tex is None else np.ascontiguousarray(tex, np.float32).reshape(nv, 3)
        tri = np.ascontiguousarray(triangles, np.uint32).reshape(-1, 3)
        mat = None if tri_material is None else np.ascontiguousarray(tri_material, np.uint32).reshape(tri.shape[0])
        h = C.c_void_p()
        _lib.check(
            _lib.lib().mp_scene_from_triangles_mat(
                ctx.handle if ctx else None, pos.ctypes.data, nrm.ctypes.data if nrm is not None else None,
                tx.ctypes.data if tx is not None else None, nv, tri.ctypes.data, mat.ctypes.data if mat is not None else None,
                tri.shape[0], C.byref(h),
            )
        )
        return cls(h, ctx)

    @classmethod
    def from_arrays(cls, inner, packets, shading, vertex_normals, vertex_tex, root_link, bbox_min, bbox_max,
                    ctx: Optional[Context] = None, tri_material=None) -> "TriangleBvh":
        """mp_scene_from_arrays: a TriangleBvh the caller already holds, in the reference's own layout
        (triangle_bvh/mod.rs:20-53) -- the inverse of export().  Nothing is rebuilt: triangle_index values and the lane
        order inside leaves are the caller's."""
        inner = np.ascontiguousarray(inner, np.uint8).reshape(-1, 128)
        packets = np.ascontiguousarray(packets, np.uint8).reshape(-1, 144)
        shading = np.ascontiguousarray(shading, np.uint32).reshape(-1, 4)
        vn = np.ascontiguousarray(vertex_normals, np.float32).reshape(-1, 3)
        vt = None if vertex_tex is None else np.ascontiguousarray(vertex_tex, np.float32).reshape(-1, 3)
        mat = None if tri_material is None else np.ascontiguousarray(tri_material, np.uint32).reshape(-1)
        # the C side copies packet_count * 8 rows of each per-triangle array: a short array would be read out of bounds
        if shading.shape[0] != packets.shape[0] * 8:
            raise ValueError(f"shading must have packet_count * 8 = {packets.shape[0] * 8} rows, got {shading.shape[0]}")
        if mat is not None and mat.shape[0] != packets.shape[0] * 8:
            raise ValueError(f"tri_material must have packet_count * 8 = {packets.shape[0] * 8} entries, got {mat.shape[0]}")
        if vt is not None and vt.shape[0] != vn.shape[0]:
            raise ValueError("vertex_tex must have one row per vertex normal")
        d = _lib.BvhDesc(
            inner.ctypes.data, packets.ctypes.data, shading.ctypes.data, mat.ctypes.data if mat is not None else None,
            vn.ctypes.data, vt.ctypes.data if vt is not None else None, inner.shape[0], packets.shape[0], vn.shape[0],
            int(root_link), (C.c_float * 3)(*[float(x) for x in bbox_min]), (C.c_float * 3)(*[float(x) for x in bbox_max]),
        )
        h = C.c_void_p()
        _lib.check(_lib.lib().mp_scene_from_arrays(ctx.handle if ctx else None, C.byref(d), C.byref(h)))
        return cls(h, ctx)

    def set_materials(self, table, sky: float = 1.0) -> None:
        """Material table and sky radiance of the build-defined path extension.  Entries: (albedo, emission) with scalars (grey)
        or (r, g, b) triples, or a dict {"albedo", "emission", "albedo2", "checker": cells per unit of texture coordinate} -- a
        procedural checkerboard over HitRecord.texture_coords (geometry/mod.rs:78-79)."""
        t = [_lib.Material.make(e) for e in table]
        arr = (_lib.Material * max(len(t), 1))(*t)
        _lib.check(_lib.lib().mp_scene_set_materials(self.handle, arr, len(t), C.c_float(sky)))

    def material_name(self, i: int) -> Optional[str]:
        n = _lib.lib().mp_scene_material_name(self.handle, int(i))
        return None if n is None else n.decode()

    def info(self) -> _lib.SceneInfo:
        out = _lib.SceneInfo()
        _lib.check(_lib.lib().mp_scene_info_get(self.handle, C.byref(out)))
        return out

    def get_bounding_box(self):
        """Object::get_bounding_box (scene/mod.rs:9)."""
        i = self.info()
        return np.array(list(i.bbox_min), np.float32), np.array(list(i.bbox_max), np.float32)

    def device_tree(self, literal: bool = False):
        """The traversal-format node array the kernels walk (mp_scene_device_tree; diagnostics / tests): returns
        (nodes (n, 8, 8) u32 view of {min.xyz, max.xyz, link, n} records, root link, stack bound, absorbed reference nodes).
        literal=False: the wide tree (thin nodes absorbed into their parents); True: the literal reference tree."""
        n, root, bound, absorbed = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        which = 1 if literal else 0
        _lib.check(_lib.lib().mp_scene_device_tree(self.handle, which, None, C.byref(n), C.byref(root), C.byref(bound), C.byref(absorbed)))
        nodes = np.zeros((n.value, 8, 8), np.uint32)
        _lib.check(_lib.lib().mp_scene_device_tree(self.handle, which, nodes.ctypes.data, None, None, None, None))
        return nodes, root.value, bound.value, absorbed.value

    def export(self, with_material: bool = False):
        """Reference-layout arrays: inner nodes (n,128) u8, packets (n,144) u8, tri shading (n*8,4) u32,
        vertex normals / tex (nv,3) f32 (+ material id per triangle slot (n*8,) u32 with with_material=True)."""
        i = self.info()
        inner = np.zeros((i.inner_count, 128), np.uint8)
        packets = np.zeros((i.packet_count, 144), np.uint8)
        shading = np.zeros((i.packet_count * 8, 4), np.uint32)
        vn = np.zeros((i.vertex_count, 3), np.float32)
        vt = np.zeros((i.vertex_count, 3), np.float32)
        mat = np.zeros((i.packet_count * 8,), np.uint32)
        _lib.check(
            _lib.lib().mp_scene_export(
                self.handle, inner.ctypes.data, packets.ctypes.data, shading.ctypes.data, vn.ctypes.data, vt.ctypes.data,
                mat.ctypes.data,
            )
        )
        return (inner, packets, shading, vn, vt, mat) if with_material else (inner, packets, shading, vn, vt)

    def intersect(self, origins, directions, stream=None, full: bool = False):
        """impl Object for TriangleBvh::intersect (ray_bvh_intersection.rs:26-96), batched over CUDA/HIP tensors.

        origins, directions: float32 torch tensors [n,3] on this context's GPU (directions need not be unit).
        Returns dict of device tensors: t, prim, u, v (+ point, normal, tex when full=True)."""
        import torch

        if self.ctx is None:
            raise _lib.MinipathError(1, "host-only BVH cannot be traced: pass a Context to with_obj/build")
        assert origins.is_cuda and directions.is_cuda and origins.dtype == torch.float32
        n = origins.shape[0]
        o = origins.t().contiguous()
        d = directions.t().contiguous()
        dev = origins.device
        out = {
            "t": torch.empty(n, dtype=torch.float32, device=dev),
            "prim": torch.empty(n, dtype=torch.int32, device=dev),
            "u": torch.empty(n, dtype=torch.float32, device=dev),
            "v": torch.empty(n, dtype=torch.float32, device=dev),
        }
        hits = _lib.HitsSoA(out["t"].data_ptr(), out["prim"].data_ptr(), out["u"].data_ptr(), out["v"].data_ptr(), None, None, None, None, None)
        if full:
            for k in ("point", "normal", "tex"):
                out[k] = torch.empty((n, 3), dtype=torch.float32, device=dev)
            out["material"] = torch.empty(n, dtype=torch.int32, device=dev)
            hits.d_point, hits.d_normal, hits.d_tex = out["point"].data_ptr(), out["normal"].data_ptr(), out["tex"].data_ptr()
            hits.d_material = out["material"].data_ptr()
            out["instance"] = torch.empty(n, dtype=torch.int32, device=dev)
            hits.d_instance = out["instance"].data_ptr()
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        _lib.check(
            _lib.lib().mp_trace_rays(
                self.ctx.handle, self.handle, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), d[0].data_ptr(),
                d[1].data_ptr(), d[2].data_ptr(), n, C.byref(hits), C.c_void_p(st),
            )
        )
        return out

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().mp_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Instances(TriangleBvh):
    """BUILD-DEFINED Object (scene/mod.rs:7-10 `trait Object`; the reference's Scene holds one object, no transforms): translated
    instances of one TriangleBvh (mp_scene_instances).  Shares the object's device arrays: keeps a reference to it."""

    def __init__(self, obj: TriangleBvh, translations):
        t = np.ascontiguousarray(translations, np.float32).reshape(-1, 3)
        h = C.c_void_p()
        _lib.check(_lib.lib().mp_scene_instances(obj.ctx.handle if obj.ctx else None, obj.handle, t.ctypes.data, t.shape[0], C.byref(h)))
        super().__init__(h, obj.ctx)
        self.object, self.translations = obj, t

    def close(self):
        super().close()  # before the object it borrows from
        self.object = None


class ObjectGroup(TriangleBvh):
    """BUILD-DEFINED Object: a top-level list of members {object, translation} over any TriangleBvh or Sphere scenes of one
    context (mp_scene_group); hits carry the member index (`instance`) and the triangle index inside that member.  Shares the members'
    device arrays: keeps references to them."""

    def __init__(self, objects, translations, rotations=None):
        """rotations: optional unit quaternions (i, j, k, w), one per member: world = q * local + translation."""
        objects = list(objects)
        t = np.ascontiguousarray(translations, np.float32).reshape(-1, 3)
        if len(objects) != t.shape[0] or not objects:
            raise ValueError("one translation per member")
        q = None
        if rotations is not None:
            q = np.ascontiguousarray(rotations, np.float32).reshape(-1, 4)
            if q.shape[0] != t.shape[0]:
                raise ValueError("one rotation per member")
        ctx = objects[0].ctx
        arr = (C.c_void_p * len(objects))(*[o.handle for o in objects])
        h = C.c_void_p()
        _lib.check(_lib.lib().mp_scene_group(ctx.handle if ctx else None, arr, q.ctypes.data if q is not None else None, t.ctypes.data,
                                             len(objects), C.byref(h)))
        super().__init__(h, ctx)
        self.objects, self.translations, self.rotations = objects, t, q

    def close(self):
        super().close()  # before the members it borrows from
        self.objects = []


class Sphere(TriangleBvh):
    """scene/primitives.rs:10-56: analytic sphere as the scene's Object (same handle type, no BVH arrays)."""

    def __init__(self, center, radius: float, ctx: Optional[Context] = None):
        h = C.c_void_p()
        c = (C.c_float * 3)(*[float(v) for v in center])
        _lib.check(_lib.lib().mp_scene_sphere(ctx.handle if ctx else None, c, C.c_float(radius), C.byref(h)))
        super().__init__(h, ctx)
        self.center, self.radius = tuple(float(v) for v in center), float(radius)


class Scene:
    """scene/mod.rs:12-15: Scene { object }."""

    def __init__(self, object: TriangleBvh):
        self.object = object
