"""Mirror of src/screen_block.rs: ScreenBlock = AABB<Point2<u32>> and its tile ordering."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Iterator, List

from . import _lib


@dataclass(frozen=True)
class ScreenBlock:
    min_x: int
    min_y: int
    max_x: int
    max_y: int

    @classmethod
    def with_size(cls, x: int, y: int, w: int, h: int) -> "ScreenBlock":
        return cls(x, y, x + w, y + h)

    def is_empty(self) -> bool:  # screen_block.rs:10-12
        return not (self.min_x < self.max_x and self.min_y < self.max_y)

    def width(self) -> int:
        return self.max_x - self.min_x

    def height(self) -> int:
        return self.max_y - self.min_y

    def area(self) -> int:  # :14-20
        return 0 if self.is_empty() else self.width() * self.height()

    def contains(self, x: int, y: int) -> bool:  # :22-24
        return self.min_x <= x < self.max_x and self.min_y <= y < self.max_y

    def internal_points(self) -> Iterator[tuple]:  # :28-39 (C order: x fastest)
        if self.is_empty():
            return
        for y in range(self.min_y, self.max_y):
            for x in range(self.min_x, self.max_x):
                yield (x, y)

    def tile_ordering(self, tile_size: int, shuffle_seed: int = 0) -> List["ScreenBlock"]:  # :46-81
        return tile_ordering(self, tile_size, shuffle_seed)

    def as_struct(self) -> _lib.Block:
        return _lib.Block(self.min_x, self.min_y, self.max_x, self.max_y)


def tile_ordering(block: ScreenBlock, tile_size: int, shuffle_seed: int = 0) -> List[ScreenBlock]:
    """screen_block.rs:46-81 through the C ABI.  shuffle_seed 0 = deterministic row-major grid."""
    if tile_size <= 0:
        raise ValueError("tile_size is a NonZeroU32")
    n = C.c_size_t(0)
    L = _lib.lib()
    _lib.check(L.mp_tile_ordering(block.as_struct(), tile_size, C.c_uint64(shuffle_seed), None, 0, C.byref(n)))
    arr = (_lib.Block * max(n.value, 1))()
    _lib.check(L.mp_tile_ordering(block.as_struct(), tile_size, C.c_uint64(shuffle_seed), arr, n.value, C.byref(n)))
    return [ScreenBlock(*arr[i].as_tuple()) for i in range(n.value)]
