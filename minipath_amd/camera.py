"""Mirror of src/camera.rs: Camera (builder-style, immutable) and CameraSampler."""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, replace
from typing import Sequence

import numpy as np

from . import _lib


def _v3(v: Sequence[float]):
    return (C.c_float * 3)(float(v[0]), float(v[1]), float(v[2]))


@dataclass(frozen=True)
class CameraSampler:
    """camera.rs:26-39 -- the 15 floats handed to the kernels."""

    values: tuple

    def as_struct(self) -> _lib.SamplerStruct:
        s = _lib.SamplerStruct()
        a = np.asarray(self.values, np.float32)
        C.memmove(C.byref(s), a.ctypes.data, 60)
        return s

    def as_array(self) -> np.ndarray:
        return np.asarray(self.values, np.float32)

    @classmethod
    def from_struct(cls, s: _lib.SamplerStruct) -> "CameraSampler":
        return cls(tuple(np.frombuffer(bytes(s), np.float32).tolist()))


@dataclass(frozen=True)
class Camera:
    """camera.rs:9-18.  camera_to_world is (unit quaternion i,j,k,w ; translation)."""

    q: tuple = (0.0, 0.0, 0.0, 1.0)
    t: tuple = (0.0, 0.0, 0.0)
    focus_distance_: float = math.inf
    sensor_is_width: bool = False
    sensor_size: float = 24e-3
    focal_length: float = 50e-3
    f_number_: float = 9.0

    @classmethod
    def default(cls) -> "Camera":
        """camera.rs:42-52: 35 mm sensor height, 50 mm f/9, looks along -Z, focused at infinity."""
        return cls()

    def _struct(self) -> _lib.CameraStruct:
        c = _lib.CameraStruct()
        c.q[:] = self.q
        c.t[:] = self.t
        c.focus_distance = self.focus_distance_
        c.sensor_is_width = 1 if self.sensor_is_width else 0
        c.sensor_size = self.sensor_size
        c.focal_length = self.focal_length
        c.f_number = self.f_number_
        return c

    def _from_struct(self, c: _lib.CameraStruct) -> "Camera":
        return replace(self, q=tuple(c.q), t=tuple(c.t), focus_distance_=float(c.focus_distance))

    def focus_distance(self, d: float) -> "Camera":
        assert d >= 0.0  # camera.rs:64
        return replace(self, focus_distance_=float(d))

    def sensor_width(self, w: float) -> "Camera":
        assert w > 0.0
        return replace(self, sensor_is_width=True, sensor_size=float(w))

    def sensor_height(self, h: float) -> "Camera":
        assert h > 0.0
        return replace(self, sensor_is_width=False, sensor_size=float(h))

    def f_number(self, n: float) -> "Camera":
        assert n > 0.0
        return replace(self, f_number_=float(n))

    def look_at(self, center, look_at, up) -> "Camera":
        """camera.rs:93-101: also focuses at `look_at`."""
        c = self._struct()
        _lib.check(_lib.lib().mp_camera_look_at(C.byref(c), _v3(center), _v3(look_at), _v3(up)))
        return self._from_struct(c)

    def look_direction(self, center, forward, up) -> "Camera":
        """camera.rs:104-116."""
        c = self._struct()
        _lib.check(_lib.lib().mp_camera_look_direction(C.byref(c), _v3(center), _v3(forward), _v3(up)))
        return replace(self, q=tuple(c.q), t=tuple(c.t))

    def translated(self, t) -> "Camera":
        """Camera::transformed with a Translation3 (camera.rs:119-121)."""
        c = self._struct()
        _lib.check(_lib.lib().mp_camera_translate(C.byref(c), _v3(t)))
        return replace(self, t=tuple(c.t))

    def center_forward_up_right(self):
        """camera.rs:148-171."""
        c = self._struct()
        out = [(C.c_float * 3)() for _ in range(4)]
        _lib.check(_lib.lib().mp_camera_basis(C.byref(c), *out))
        return tuple(np.array(list(o), np.float32) for o in out)

    def build_sampler(self, resolution) -> CameraSampler:
        """camera.rs:123-146."""
        c = self._struct()
        s = _lib.SamplerStruct()
        _lib.check(_lib.lib().mp_camera_build_sampler(C.byref(c), int(resolution[0]), int(resolution[1]), C.byref(s)))
        return CameraSampler.from_struct(s)

    @classmethod
    def teapot_view(cls) -> "Camera":
        """benches/render_teapot.rs:12-19."""
        return cls.default().look_at((0.0, 2.0, 10.0), (0.0, 1.5, 0.0), (0.0, 1.0, 0.0)).f_number(4.8).focus_distance(10.0)
