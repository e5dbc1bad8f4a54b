"""`pixo::jpeg` — host-side mirror of the reference's JPEG API over the HIP C ABI.

Same names, argument meaning and error behaviour as the reference
(src/jpeg/mod.rs:88-447, src/wasm.rs:113-142):

    encode(data, options) -> bytes                      jpeg/mod.rs:88
    encode_into(output: bytearray, data, options)       jpeg/mod.rs:328
    JpegOptions / JpegOptions.builder / presets         jpeg/mod.rs:121-300
    Subsampling                                         jpeg/mod.rs:96-101
    encode_jpeg(data, w, h, color_type, quality, preset, subsampling_420)   wasm.rs:113

plus the device seam (the reference's internal `YCbCrCoefficients`, jpeg/mod.rs:58-61):

    coefficients(data, options) -> (y, cb, cr) int16 arrays [blocks, 64]
    coefficients_device(...)    -> same on device pointers / torch tensors, async

and one image over several GPUs: `BandEncoder`, `splice`, `encode_multi`, host twins
`band_*_host`; `sharded.py` drives them over `torch.distributed`.
"""
import ctypes as C
import dataclasses
import enum
from typing import Optional

import numpy as np

from . import _lib
from .color import ColorType
from . import error
from .error import from_status


class Subsampling(enum.IntEnum):
    S444 = 0
    S420 = 1


@dataclasses.dataclass
class JpegOptions:
    """jpeg/mod.rs:121-157 (Default: quality 75, 4:4:4, RGB, width/height 0)."""
    width: int = 0
    height: int = 0
    color_type: ColorType = ColorType.Rgb
    quality: int = 75
    subsampling: Subsampling = Subsampling.S444
    restart_interval: Optional[int] = None
    optimize_huffman: bool = False
    progressive: bool = False
    trellis_quant: bool = False

    # presets, jpeg/mod.rs:162-216
    @staticmethod
    def fast(width, height, quality):
        return JpegOptions(width, height, ColorType.Rgb, quality, Subsampling.S444)

    @staticmethod
    def balanced(width, height, quality):
        return JpegOptions(width, height, ColorType.Rgb, quality, Subsampling.S444, None, True)

    @staticmethod
    def max(width, height, quality):
        return JpegOptions(width, height, ColorType.Rgb, quality, Subsampling.S420, None, True, True, True)

    @staticmethod
    def from_preset(width, height, quality, preset):
        if preset == 0:
            return JpegOptions.fast(width, height, quality)
        if preset == 2:
            return JpegOptions.max(width, height, quality)
        return JpegOptions.balanced(width, height, quality)

    @staticmethod
    def builder(width, height):
        return JpegOptionsBuilder(width, height)

    def _c(self) -> _lib.JpegOptionsC:
        o = _lib.JpegOptionsC()
        o.width, o.height = self.width & 0xFFFFFFFF, self.height & 0xFFFFFFFF
        o.color_type, o.quality = int(self.color_type) & 0xFF, int(self.quality) & 0xFF
        o.subsampling = int(self.subsampling)
        o.has_restart_interval = 0 if self.restart_interval is None else 1
        o.restart_interval = 0 if self.restart_interval is None else int(self.restart_interval) & 0xFFFF
        o.optimize_huffman = int(bool(self.optimize_huffman))
        o.progressive = int(bool(self.progressive))
        o.trellis_quant = int(bool(self.trellis_quant))
        return o


class JpegOptionsBuilder:
    """jpeg/mod.rs:230-300."""

    def __init__(self, width, height):
        self._o = JpegOptions(width=width, height=height, color_type=ColorType.Rgb)

    def color_type(self, color_type):
        self._o.color_type = color_type
        return self

    def quality(self, quality):
        self._o.quality = quality
        return self

    def subsampling(self, subsampling):
        self._o.subsampling = subsampling
        return self

    def restart_interval(self, interval):
        self._o.restart_interval = interval
        return self

    def optimize_huffman(self, value):
        self._o.optimize_huffman = value
        return self

    def progressive(self, value):
        self._o.progressive = value
        return self

    def trellis_quant(self, value):
        self._o.trellis_quant = value
        return self

    def preset(self, preset):
        """Applies a preset while retaining dimensions, colour type and quality (:285-293)."""
        keep = self._o
        self._o = JpegOptions.from_preset(keep.width, keep.height, keep.quality, preset)
        self._o.color_type = keep.color_type
        return self

    def build(self):
        return dataclasses.replace(self._o)


def _raise(status):
    raise from_status(status, _lib.load().pixo_hip_last_error().decode())


def _as_u8(data):
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    if a.dtype != np.uint8:
        raise TypeError("pixel data must be bytes-like / uint8")
    return np.ascontiguousarray(a).reshape(-1)


def encode(data, options: JpegOptions) -> bytes:
    """`pixo::jpeg::encode` (jpeg/mod.rs:88): complete JFIF file as bytes."""
    L = _lib.load()
    px = _as_u8(data)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    oc = options._c()
    rc = L.pixo_hip_jpeg_encode(px.ctypes.data, px.size, C.byref(oc), C.byref(out), C.byref(n))
    if rc:
        _raise(rc)
    try:
        return _lib.file_bytes(L, out, n.value)
    finally:
        L.pixo_hip_free(out)


def encode_into(output: bytearray, data, options: JpegOptions) -> None:
    """`pixo::jpeg::encode_into` (jpeg/mod.rs:328): clears and refills `output`; on error
    `output` is left untouched (validation precedes `output.clear()` in the reference)."""
    blob = encode(data, options)
    del output[:]
    output += blob


def encode_into_buffer(buffer: np.ndarray, data, options: JpegOptions) -> int:
    """The fixed-capacity form of `encode_into` (`pixo_hip_jpeg_encode_into`): writes the file into the
    caller's uint8 array and returns its length.  Raises `error.BufferTooSmall` (its `.needed` says how
    many bytes the file has) when it does not fit — the reserve-and-retry protocol a Rust `Vec` would use.  What `buffer`
    holds after that error is unspecified: pageable memory is left alone, but a PINNED buffer is written by the GPU directly,
    up to its capacity and never beyond (include/pixo_hip.h)."""
    L = _lib.load()
    px = _as_u8(data)
    if buffer.dtype != np.uint8 or not buffer.flags["C_CONTIGUOUS"]:
        raise TypeError("buffer must be a contiguous uint8 array")
    n = C.c_size_t()
    oc = options._c()
    rc = L.pixo_hip_jpeg_encode_into(buffer.ctypes.data, buffer.size, px.ctypes.data, px.size, C.byref(oc), C.byref(n))
    if rc:
        try:
            _raise(rc)
        except error.BufferTooSmall as e:
            e.needed = n.value
            raise
    return n.value


def encode_jpeg(data, width, height, color_type, quality, preset, subsampling_420) -> bytes:
    """The reference's flat wasm export `encode_jpeg` (src/wasm.rs:113-142), same 7 arguments."""
    L = _lib.load()
    px = _as_u8(data)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    rc = L.pixo_hip_encode_jpeg(px.ctypes.data, px.size, width, height, color_type & 0xFF,
                                quality & 0xFF, preset & 0xFF, int(bool(subsampling_420)),
                                C.byref(out), C.byref(n))
    if rc:
        _raise(rc)
    try:
        return _lib.file_bytes(L, out, n.value)
    finally:
        L.pixo_hip_free(out)


def coefficient_geometry(width, height, color_type=ColorType.Rgb, subsampling=Subsampling.S420):
    """(y_blocks, c_blocks) of the coefficient tuple."""
    L = _lib.load()
    yb, cb = C.c_size_t(), C.c_size_t()
    rc = L.pixo_hip_coeff_geometry(width, height, int(color_type), int(subsampling), C.byref(yb), C.byref(cb))
    if rc:
        _raise(rc)
    return yb.value, cb.value


def coefficients(data, options: JpegOptions):
    """The GPU half only: quantised DCT blocks in the reference's `YCbCrCoefficients`
    layout (natural order; 4:2:0 stores Y as 4 blocks per MCU)."""
    L = _lib.load()
    px = _as_u8(data)
    expected = options.width * options.height * ColorType(options.color_type).bytes_per_pixel() \
        if options.color_type in (0, 2) else -1
    yb, cbn = coefficient_geometry(options.width, options.height, options.color_type, options.subsampling)
    if expected >= 0 and px.size != expected:
        raise from_status(-2, "Invalid pixel data length: expected %d bytes, got %d" % (expected, px.size))
    y = np.empty((yb, 64), np.int16)
    cb = np.empty((cbn, 64), np.int16)
    cr = np.empty((cbn, 64), np.int16)
    rc = L.pixo_hip_jpeg_coeffs(px.ctypes.data, options.width, options.height, int(options.color_type),
                                int(options.subsampling), int(options.quality), y.ctypes.data, yb,
                                cb.ctypes.data, cr.ctypes.data, cbn)
    if rc:
        _raise(rc)
    return y, cb, cr


def coefficients_device(d_pixels, width, height, color_type, subsampling, quality, d_y, d_cb, d_cr,
                        batch=1, stream=0):
    """Asynchronous launch on device memory.  Pointers may be ints or objects with
    `.data_ptr()` (torch tensors); `stream` is a hipStream_t handle (int), 0 = default."""
    L = _lib.load()

    def ptr(x):
        if x is None:
            return None
        return x.data_ptr() if hasattr(x, "data_ptr") else int(x)

    rc = L.pixo_hip_jpeg_coeffs_device(ptr(d_pixels), width, height, int(color_type), int(subsampling),
                                       int(quality), batch, ptr(d_y), ptr(d_cb), ptr(d_cr),
                                       C.c_void_p(stream) if stream else None)
    if rc:
        _raise(rc)


def coefficients_integer(data, options: JpegOptions):
    """The INTEGER secondary mode (`pixo_hip_jpeg_coeffs_integer`): the reference's fixed-point DCT
    family — dead code in its `encode()` — per 8x8 block, 4:4:4 RGB or gray.  Never used by `encode`."""
    L = _lib.load()
    px = _as_u8(data)
    yb, cbn = coefficient_geometry(options.width, options.height, options.color_type, Subsampling.S444)
    expected = options.width * options.height * (1 if int(options.color_type) == 0 else 3)
    if px.size != expected:
        raise from_status(-2, "Invalid pixel data length: expected %d bytes, got %d" % (expected, px.size))
    y = np.empty((yb, 64), np.int16)
    cb = np.empty((cbn, 64), np.int16)
    cr = np.empty((cbn, 64), np.int16)
    rc = L.pixo_hip_jpeg_coeffs_integer(px.ctypes.data, options.width, options.height, int(options.color_type),
                                        int(options.subsampling), int(options.quality), y.ctypes.data, yb,
                                        cb.ctypes.data, cr.ctypes.data, cbn)
    if rc:
        _raise(rc)
    return y, cb, cr


def entropy_encode(y, cb, cr, options: JpegOptions) -> bytes:
    """Host entropy stage from an existing coefficient tuple (used when several GPUs each
    produced a band of it)."""
    L = _lib.load()
    y = np.ascontiguousarray(y, np.int16)
    cb = np.ascontiguousarray(cb, np.int16)
    cr = np.ascontiguousarray(cr, np.int16)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    oc = options._c()
    rc = L.pixo_hip_jpeg_entropy_encode(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, C.byref(oc),
                                        C.byref(out), C.byref(n))
    if rc:
        _raise(rc)
    try:
        return _lib.file_bytes(L, out, n.value)
    finally:
        L.pixo_hip_free(out)


def _dev_ptr(x):
    if x is None:
        return None
    return x.data_ptr() if hasattr(x, "data_ptr") else int(x)


def entropy_encode_device(d_y, d_cb, d_cr, options: JpegOptions) -> bytes:
    """Device entropy stage: coefficient tuple in HBM (torch tensors or raw pointers, one image)
    -> JPEG file.  Per-block Huffman coding, bit-offset scan, packing and 0xFF stuffing run on
    the GPU; only the file crosses PCIe."""
    L = _lib.load()
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    oc = options._c()
    rc = L.pixo_hip_jpeg_entropy_encode_device(_dev_ptr(d_y), _dev_ptr(d_cb), _dev_ptr(d_cr), C.byref(oc),
                                               C.byref(out), C.byref(n))
    if rc:
        _raise(rc)
    try:
        return _lib.file_bytes(L, out, n.value)
    finally:
        L.pixo_hip_free(out)


def encode_device(d_pixels, options: JpegOptions) -> bytes:
    """`encode` for pixels that are already in HBM (torch tensor or raw pointer)."""
    L = _lib.load()
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    oc = options._c()
    rc = L.pixo_hip_jpeg_encode_device(_dev_ptr(d_pixels), C.byref(oc), C.byref(out), C.byref(n))
    if rc:
        _raise(rc)
    try:
        return _lib.file_bytes(L, out, n.value)
    finally:
        L.pixo_hip_free(out)


def encode_device_into(buffer, d_pixels, options: JpegOptions) -> int:
    """Device pixels -> file written straight into `buffer` (a contiguous uint8 numpy array or a torch CPU
    tensor, ideally pinned: then the device-to-host copy is the only pass over the file).  Returns the
    file's length; raises `error.BufferTooSmall` (with `.needed`) when it does not fit — a pinned `buffer` may then have been
    written up to its capacity (small files are stored into pinned memory by the stuffing kernel itself), never beyond it."""
    L = _lib.load()
    if hasattr(buffer, "data_ptr"):
        ptr, cap = buffer.data_ptr(), buffer.numel() * buffer.element_size()
    else:
        ptr, cap = buffer.ctypes.data, buffer.nbytes
    n = C.c_size_t()
    oc = options._c()
    rc = L.pixo_hip_jpeg_encode_device_into(_dev_ptr(d_pixels), C.byref(oc), ptr, cap, C.byref(n))
    if rc:
        try:
            _raise(rc)
        except error.BufferTooSmall as e:
            e.needed = n.value
            raise
    return n.value


def encode_batch_device(d_pixels, options: JpegOptions, batch: int):
    """`batch` equally sized images back to back in HBM -> list of `batch` JPEG files; one
    coefficient launch and one pass of the device entropy stage for all of them."""
    L = _lib.load()
    files = (C.POINTER(C.c_uint8) * batch)()
    lens = (C.c_size_t * batch)()
    oc = options._c()
    rc = L.pixo_hip_jpeg_encode_batch_device(_dev_ptr(d_pixels), C.byref(oc), batch, files, lens)
    if rc:
        _raise(rc)
    out = []
    for i in range(batch):
        out.append(_lib.file_bytes(L, files[i], lens[i]))
        L.pixo_hip_free(files[i])
    return out


def encode_batch_device_raw(d_pixels, options: JpegOptions, batch: int):
    """`pixo_hip_jpeg_encode_batch_device` as a C caller sees it: (array of `batch` pointers to the files, array of their lengths) —
    blocks the caller owns until `free_files`.  (No Python bytes objects: 64 x 1.4 MB of those are 22,000 page faults of their own.)"""
    L = _lib.load()
    files = (C.POINTER(C.c_uint8) * batch)()
    lens = (C.c_size_t * batch)()
    oc = options._c()
    rc = L.pixo_hip_jpeg_encode_batch_device(_dev_ptr(d_pixels), C.byref(oc), batch, files, lens)
    if rc:
        _raise(rc)
    return files, lens


def free_files(files, batch: int) -> None:
    L = _lib.load()
    for i in range(batch):
        L.pixo_hip_free(files[i])


def encode_batch_device_into(arena, d_pixels, options: JpegOptions, batch: int):
    """`pixo_hip_jpeg_encode_batch_device_into`: the `batch` files back to back in `arena` (a torch uint8 CPU tensor —
    ideally pinned —, a numpy uint8 array, or None for a size query), every file copied from the device straight to its
    final place.  Returns (offsets, lens); raises BufferTooSmall (`.needed`) when the files do not fit."""
    L = _lib.load()
    offsets = (C.c_size_t * batch)()
    lens = (C.c_size_t * batch)()
    if arena is None:
        ptr, cap = None, 0
    elif hasattr(arena, "data_ptr"):
        ptr, cap = arena.data_ptr(), arena.numel()
    else:
        ptr, cap = arena.ctypes.data, arena.size
    oc = options._c()
    rc = L.pixo_hip_jpeg_encode_batch_device_into(_dev_ptr(d_pixels), C.byref(oc), batch, ptr, cap, offsets, lens)
    if rc == -9 and arena is None:  # PIXO_ERR_BUFFER_TOO_SMALL: the answer to a size query
        return list(offsets), list(lens)
    if rc == -9:  # (offsets / lens were filled in: the size a second attempt needs)
        try:
            _raise(rc)
        except error.BufferTooSmall as e:
            e.needed = int(offsets[batch - 1] + lens[batch - 1]) if batch else 0
            raise
    if rc:
        _raise(rc)
    return list(offsets), list(lens)


def encode_batch_multi(arena, pixels, options: JpegOptions, batch: int, devices):
    """`pixo_hip_jpeg_encode_batch_multi`: `batch` equally sized images — a device tensor / pointer on any GPU of this process,
    or a numpy array in host memory — encoded by `devices` (one contiguous run of images each; a device may be listed more than
    once), every GPU copying its files to their final place in `arena` (torch uint8 CPU tensor, ideally pinned, numpy uint8
    array, or None for a size query) over its own PCIe link.  Returns (offsets, lens); raises BufferTooSmall (`.needed`)."""
    L = _lib.load()
    offsets = (C.c_size_t * batch)()
    lens = (C.c_size_t * batch)()
    if arena is None:
        ptr, cap = None, 0
    elif hasattr(arena, "data_ptr"):
        ptr, cap = arena.data_ptr(), arena.numel()
    else:
        ptr, cap = arena.ctypes.data, arena.size
    # (the C entry takes no length: it reads batch * image_bytes from the pointer — whatever carries a size is checked here)
    bpp = {int(ColorType.Gray): 1, int(ColorType.Rgb): 3}.get(int(options.color_type))
    need = batch * options.width * options.height * bpp if bpp else None
    if hasattr(pixels, "data_ptr"):
        have = pixels.numel() * pixels.element_size()
        src = pixels.data_ptr()
    elif isinstance(pixels, np.ndarray):
        pixels = _as_u8(pixels)
        have = pixels.size
        src = pixels.ctypes.data
    else:
        have = None
        src = int(pixels)
    if need is not None and have is not None and have != need:
        raise error.InvalidDataLength("Invalid pixel data length: expected %d bytes, got %d" % (need, have))
    if len(devices) == 0:
        raise error.CompressionError("Compression error: encode_batch_multi: no device listed")
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    oc = options._c()
    rc = L.pixo_hip_jpeg_encode_batch_multi(src, C.byref(oc), batch, devs, len(devices), ptr, cap, offsets, lens)
    if rc == -9 and arena is None:  # PIXO_ERR_BUFFER_TOO_SMALL: the answer to a size query
        return list(offsets), list(lens)
    if rc == -9:
        try:
            _raise(rc)
        except error.BufferTooSmall as e:
            e.needed = int(offsets[batch - 1] + lens[batch - 1]) if batch else 0
            raise
    if rc:
        _raise(rc)
    return list(offsets), list(lens)


def debug_stream_copy(d_in, d_out, nbytes, stream=0):
    """MEASUREMENT only (`pixo_hip_debug_stream_copy`): a plain device copy in the coefficient kernel's launch shape."""
    def ptr(x):
        return x.data_ptr() if hasattr(x, "data_ptr") else int(x)
    rc = _lib.load().pixo_hip_debug_stream_copy(ptr(d_in), ptr(d_out), int(nbytes), C.c_void_p(stream) if stream else None)
    if rc:
        _raise(rc)


def debug_engine_clock(stream=0) -> float:
    """MEASUREMENT only (`pixo_hip_debug_engine_clock`): the engine clock in Hz under full vector load."""
    hz = C.c_double(0.0)
    rc = _lib.load().pixo_hip_debug_engine_clock(C.c_void_p(stream) if stream else None, C.byref(hz))
    if rc:
        _raise(rc)
    return hz.value


def debug_stream_io(d_in, d_out, workgroups, loads, stores, stream=0):
    """MEASUREMENT only (`pixo_hip_debug_stream_io`): `workgroups` x 192 threads, each `loads` 16-byte loads then `stores` 16-byte stores."""
    def ptr(x):
        return x.data_ptr() if hasattr(x, "data_ptr") else int(x)
    rc = _lib.load().pixo_hip_debug_stream_io(ptr(d_in), ptr(d_out), int(workgroups), int(loads), int(stores), C.c_void_p(stream) if stream else None)
    if rc:
        _raise(rc)


def debug_scan_device_async(d_pixels, options: JpegOptions, stream=0, batch=1) -> int:
    """MEASUREMENT only (`pixo_hip_debug_scan_device_async[_batch]`): the device kernels of one baseline file — or of `batch`
    equally sized images back to back — enqueued on `stream`, not waited for, nothing delivered.  Returns 1 when the fused
    pixel -> scan kernel ran, 0 for the two-kernel form."""
    form = C.c_int(0)
    oc = options._c()
    L = _lib.load()
    if batch == 1:
        rc = L.pixo_hip_debug_scan_device_async(_dev_ptr(d_pixels), C.byref(oc), C.c_void_p(stream) if stream else None, C.byref(form))
    else:
        rc = L.pixo_hip_debug_scan_device_async_batch(_dev_ptr(d_pixels), C.byref(oc), int(batch), C.c_void_p(stream) if stream else None, C.byref(form))
    if rc:
        _raise(rc)
    return form.value


def lookback_fallbacks() -> int:
    """How often a single-pass entropy kernel gave up waiting and the multi-pass kernels coded the scan instead (tests)."""
    return int(_lib.load().pixo_hip_debug_lookback_fallbacks())


def dispatch_gate_stats():
    """(waits, timeouts) of the single-pass kernels' dispatch gate (include/pixo_hip.h pixo_hip_debug_dispatch_gate; tests, tools)."""
    w, t = C.c_uint64(0), C.c_uint64(0)
    _lib.load().pixo_hip_debug_dispatch_gate(C.byref(w), C.byref(t))
    return int(w.value), int(t.value)


def band(width, height, color_type, subsampling, parts, index):
    """MCU-row band `index` of `parts`: dict(row_begin,row_end,y_offset,y_blocks,
    c_offset,c_blocks).  Bands are independent sub-images of the same width."""
    L = _lib.load()
    r0, r1 = C.c_uint32(), C.c_uint32()
    yo, yb, co, cbk = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    rc = L.pixo_hip_band(width, height, int(color_type), int(subsampling), parts, index,
                         C.byref(r0), C.byref(r1), C.byref(yo), C.byref(yb), C.byref(co), C.byref(cbk))
    if rc:
        _raise(rc)
    return dict(row_begin=r0.value, row_end=r1.value, y_offset=yo.value, y_blocks=yb.value,
                c_offset=co.value, c_blocks=cbk.value)


COUNT_WORDS = 536  # PIXO_HIP_COUNT_WORDS: [class][12 DC categories + 256 AC run/size symbols]


def _take(L, out, n):
    try:
        return _lib.file_bytes(L, out, n.value)
    finally:
        L.pixo_hip_free(out)


def _dc3(values):
    return (C.c_int16 * 3)(*[int(v) for v in values])


def _counts(total_counts):
    if total_counts is None:
        return None
    a = np.ascontiguousarray(total_counts, np.uint64)
    assert a.size == COUNT_WORDS
    return a


class BandEncoder:
    """One MCU-row band of an image on one GPU (`pixo_hip_band_encoder`): coefficient kernel,
    then — with what the other bands exchanged — statistics, bit length, and the packed + stuffed piece.
    The exchanges themselves (3 x i16 and one u64 per band) are the caller's: see `sharded.encode_banded`."""

    def __init__(self, options: JpegOptions, parts: int, index: int, device: int = 0):
        L = _lib.load()
        self._L = L
        self._h = C.c_void_p()
        oc = options._c()
        rc = L.pixo_hip_band_encoder_create(C.byref(oc), parts, index, device, C.byref(self._h))
        if rc:
            _raise(rc)
        r0, r1 = C.c_uint32(), C.c_uint32()
        L.pixo_hip_band_encoder_rows(self._h, C.byref(r0), C.byref(r1))
        self.row_begin, self.row_end = r0.value, r1.value

    def close(self):
        if self._h:
            self._L.pixo_hip_band_encoder_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def coeffs(self, band_pixels):
        """`band_pixels`: this band's rows — host bytes / uint8 array, or a device tensor / pointer.
        Returns the band's last DCs (Y, Cb, Cr)."""
        last = (C.c_int16 * 3)()
        if hasattr(band_pixels, "data_ptr") and getattr(band_pixels, "is_cuda", False):
            rc = self._L.pixo_hip_band_encoder_coeffs(self._h, band_pixels.data_ptr(), 1, last)
        elif isinstance(band_pixels, int):
            rc = self._L.pixo_hip_band_encoder_coeffs(self._h, band_pixels, 1, last)
        else:
            px = _as_u8(band_pixels)
            rc = self._L.pixo_hip_band_encoder_coeffs(self._h, px.ctypes.data if px.size else None, 0, last)
        if rc:
            _raise(rc)
        return [int(v) for v in last]

    def count(self, prev_dc):
        out = np.zeros(COUNT_WORDS, np.uint64)
        rc = self._L.pixo_hip_band_encoder_count(self._h, _dc3(prev_dc), out.ctypes.data_as(C.POINTER(C.c_uint64)))
        if rc:
            _raise(rc)
        return out

    def lengths(self, prev_dc, total_counts=None) -> int:
        bits = C.c_uint64()
        tc = _counts(total_counts)
        rc = self._L.pixo_hip_band_encoder_lengths(self._h, _dc3(prev_dc),
                                                   tc.ctypes.data_as(C.POINTER(C.c_uint64)) if tc is not None else None,
                                                   C.byref(bits))
        if rc:
            _raise(rc)
        return bits.value

    def pack_device(self, bit_offset: int):
        """Step 4 without the copy: returns (16-byte piece header, length of the stuffed body); the body stays
        in the encoder's device buffer until `copy_body`."""
        hdr = (C.c_uint8 * 16)()
        n = C.c_size_t()
        rc = self._L.pixo_hip_band_encoder_pack_device(self._h, bit_offset, hdr, None, C.byref(n))
        if rc:
            _raise(rc)
        return bytes(hdr), n.value

    def copy_body(self, dst) -> None:
        """The packed body -> `dst`: a torch tensor (CPU, ideally pinned, or on the encoder's GPU), a numpy
        uint8 array, or a raw address."""
        if hasattr(dst, "data_ptr"):
            ptr = dst.data_ptr()
        elif hasattr(dst, "ctypes"):
            ptr = dst.ctypes.data
        else:
            ptr = int(dst)
        rc = self._L.pixo_hip_band_encoder_copy_body(self._h, ptr)
        if rc:
            _raise(rc)

    def pack(self, bit_offset: int) -> bytes:
        out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        rc = self._L.pixo_hip_band_encoder_pack(self._h, bit_offset, C.byref(out), C.byref(n))
        if rc:
            _raise(rc)
        return _take(self._L, out, n)


def _band_host_args(y, cb, cr):
    y = np.ascontiguousarray(y, np.int16)
    cb = np.ascontiguousarray(cb, np.int16)
    cr = np.ascontiguousarray(cr, np.int16)
    return y, cb, cr


def band_count_host(y, cb, cr, options: JpegOptions, band_rows: int, prev_dc):
    """Host twin of `BandEncoder.count` on a band's tuple in host memory."""
    L = _lib.load()
    y, cb, cr = _band_host_args(y, cb, cr)
    out = np.zeros(COUNT_WORDS, np.uint64)
    oc = options._c()
    rc = L.pixo_hip_jpeg_band_count_host(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, C.byref(oc), band_rows, _dc3(prev_dc),
                                         out.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc:
        _raise(rc)
    return out


def band_bits_host(y, cb, cr, options: JpegOptions, band_rows: int, prev_dc, total_counts=None) -> int:
    L = _lib.load()
    y, cb, cr = _band_host_args(y, cb, cr)
    bits = C.c_uint64()
    tc = _counts(total_counts)
    oc = options._c()
    rc = L.pixo_hip_jpeg_band_bits_host(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, C.byref(oc), band_rows, _dc3(prev_dc),
                                        tc.ctypes.data_as(C.POINTER(C.c_uint64)) if tc is not None else None, C.byref(bits))
    if rc:
        _raise(rc)
    return bits.value


def band_piece_host(y, cb, cr, options: JpegOptions, band_rows: int, prev_dc, bit_offset: int, total_counts=None) -> bytes:
    L = _lib.load()
    y, cb, cr = _band_host_args(y, cb, cr)
    tc = _counts(total_counts)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    oc = options._c()
    rc = L.pixo_hip_jpeg_band_piece_host(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, C.byref(oc), band_rows, _dc3(prev_dc),
                                         tc.ctypes.data_as(C.POINTER(C.c_uint64)) if tc is not None else None, bit_offset,
                                         C.byref(out), C.byref(n))
    if rc:
        _raise(rc)
    return _take(L, out, n)


def splice(options: JpegOptions, pieces, total_counts=None) -> bytes:
    """Headers + the bands' pieces in order (shared bytes merged, stuffed, final 1-padding) + EOI."""
    L = _lib.load()
    n_parts = len(pieces)
    bufs = [np.frombuffer(p, np.uint8) for p in pieces]
    ptrs = (C.c_void_p * n_parts)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * n_parts)(*[b.size for b in bufs])
    tc = _counts(total_counts)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    oc = options._c()
    rc = L.pixo_hip_jpeg_splice(C.byref(oc), tc.ctypes.data_as(C.POINTER(C.c_uint64)) if tc is not None else None, ptrs, lens,
                                n_parts, C.byref(out), C.byref(n))
    if rc:
        _raise(rc)
    return _take(L, out, n)


def splice_layout(options: JpegOptions, headers, total_counts=None):
    """From the 16-byte headers of all pieces: (file length, [offset of every band's body in the file])."""
    L = _lib.load()
    n_parts = len(headers)
    blob = np.frombuffer(b"".join(h[:16] for h in headers), np.uint8)
    tc = _counts(total_counts)
    flen = C.c_size_t()
    offs = (C.c_size_t * n_parts)()
    oc = options._c()
    rc = L.pixo_hip_jpeg_splice_layout(C.byref(oc), tc.ctypes.data_as(C.POINTER(C.c_uint64)) if tc is not None else None,
                                       blob.ctypes.data, n_parts, C.byref(flen), offs)
    if rc:
        _raise(rc)
    return flen.value, list(offs)


def splice_finish(options: JpegOptions, headers, file, file_len, total_counts=None) -> None:
    """Everything of the file that is not a band's body — JFIF headers, shared bytes, padding, EOI — into
    `file` (torch CPU tensor / numpy uint8 array holding `file_len` bytes, the bodies already in place)."""
    L = _lib.load()
    blob = np.frombuffer(b"".join(h[:16] for h in headers), np.uint8)
    tc = _counts(total_counts)
    ptr = file.data_ptr() if hasattr(file, "data_ptr") else file.ctypes.data
    oc = options._c()
    rc = L.pixo_hip_jpeg_splice_finish(C.byref(oc), tc.ctypes.data_as(C.POINTER(C.c_uint64)) if tc is not None else None,
                                       blob.ctypes.data, len(headers), ptr, file_len)
    if rc:
        _raise(rc)


def encode_multi(data, options: JpegOptions, devices) -> bytes:
    """`encode` with the image's MCU-row bands spread over `devices` of this process (one host thread per
    band; a device may be listed more than once).  Byte-identical to `encode`."""
    L = _lib.load()
    px = _as_u8(data)
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    oc = options._c()
    rc = L.pixo_hip_jpeg_encode_multi(px.ctypes.data, px.size, C.byref(oc), devs, len(devices), C.byref(out), C.byref(n))
    if rc:
        _raise(rc)
    return _take(L, out, n)


def set_producer_stream(stream) -> None:
    """Device-pointer entry points are ordered after the work enqueued so far on `stream` (a hipStream_t
    handle as int, 0 / None = the NULL stream, PyTorch's default).  Per thread and sticky: see `producer_stream`."""
    _lib.load().pixo_hip_set_producer_stream(C.c_void_p(stream) if stream else None)


def get_producer_stream() -> int:
    return _lib.load().pixo_hip_get_producer_stream() or 0


class producer_stream:
    """`with producer_stream(handle): ...` — names the producer stream for the calls inside and puts the thread's
    previous setting back afterwards (the setting is per thread and sticky; a stream of another device left behind
    would make later device-pointer calls wait on the host for it)."""

    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        self.prev = get_producer_stream()
        set_producer_stream(self.stream)
        return self

    def __exit__(self, *exc):
        set_producer_stream(self.prev)
        return False


def debug_configure(switches=None) -> None:
    """Tests and A/B tools: replace the library's debug switches (`PIXO_HIP_DEBUG` syntax); None = re-read the environment."""
    _lib.load().pixo_hip_debug_configure(switches.encode() if switches is not None else None)


def device_count() -> int:
    return _lib.load().pixo_hip_device_count()


def set_device(device: int) -> None:
    rc = _lib.load().pixo_hip_set_device(device)
    if rc:
        _raise(rc)


def trim() -> None:
    """Release the calling thread's device and pinned buffers (`pixo_hip_trim`)."""
    _lib.load().pixo_hip_trim()
