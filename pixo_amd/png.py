"""PNG row filters + Adler-32 on the MI355X: the bytes the reference's
`apply_filters` (src/png/filter.rs:51-206) hands to its DEFLATE, and the zlib wrapper's checksum
of them.  Mirrors `pixo::png::FilterStrategy` (src/png/mod.rs:345-364).  No CPU fallback."""
import ctypes as C
import enum

import numpy as np

from . import _lib
from .error import from_status


def _raise(status):
    raise from_status(status, _lib.load().pixo_hip_last_error().decode())


class FilterStrategy(enum.IntEnum):
    NONE = 0
    SUB = 1
    UP = 2
    AVERAGE = 3
    PAETH = 4
    MINSUM = 5
    ADAPTIVE = 6
    ADAPTIVE_FAST = 7
    BIGRAMS = 8


NO_RAYON = 1  # flags: semantics of a reference build without the `parallel` feature


def filtered_size(width, height, bytes_per_pixel):
    return height * (width * bytes_per_pixel + 1)


def apply_filters(data, width, height, bytes_per_pixel, strategy=FilterStrategy.ADAPTIVE, flags=0):
    """Host pixels -> (filtered stream as uint8 array [height * (row_bytes + 1)], adler32)."""
    L = _lib.load()
    px = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    out = np.empty(filtered_size(width, height, bytes_per_pixel), np.uint8)
    ad = C.c_uint32()
    rc = L.pixo_hip_png_filter(px.ctypes.data, px.size, width, height, bytes_per_pixel, int(strategy), flags,
                               out.ctypes.data, out.size, C.byref(ad))
    if rc:
        _raise(rc)
    return out, ad.value


def apply_filters_device(d_data, width, height, bytes_per_pixel, d_out, strategy=FilterStrategy.ADAPTIVE, flags=0):
    """Device pixels (torch tensor / raw pointer) -> filtered stream written to d_out; returns adler32."""
    L = _lib.load()

    def ptr(x):
        return x.data_ptr() if hasattr(x, "data_ptr") else int(x)

    ad = C.c_uint32()
    rc = L.pixo_hip_png_filter_device(ptr(d_data), width, height, bytes_per_pixel, int(strategy), flags, ptr(d_out), C.byref(ad))
    if rc:
        _raise(rc)
    return ad.value


def apply_filters_async(d_data, width, height, bytes_per_pixel, d_out, d_row_sums, d_scratch,
                        strategy=FilterStrategy.ADAPTIVE, flags=0, stream=0):
    """Enqueue only (no synchronisation): d_row_sums receives 2 u64 per row; combine a host copy of
    them with `adler32_from_row_sums`."""
    L = _lib.load()

    def ptr(x):
        return x.data_ptr() if hasattr(x, "data_ptr") else int(x)

    rc = L.pixo_hip_png_filter_async(ptr(d_data), width, height, bytes_per_pixel, int(strategy), flags, ptr(d_out),
                                     ptr(d_row_sums), ptr(d_scratch), C.c_void_p(stream) if stream else None)
    if rc:
        _raise(rc)


def adler32_from_row_sums(row_sums, width, height, bytes_per_pixel):
    L = _lib.load()
    a = np.ascontiguousarray(row_sums, dtype=np.uint64)
    assert a.size == 2 * height
    return int(L.pixo_hip_png_adler32_from_row_sums(a.ctypes.data, width, height, bytes_per_pixel)) & 0xFFFFFFFF
