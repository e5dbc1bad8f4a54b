"""Loader for the C-ABI library (pixo_amd/libpixo_hip.so, built by pixo_amd/csrc/Makefile).

There is no CPU fallback: if the HIP library is missing, importing the compute entry
points fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PIXO_HIP_LIB lets kernel A/B experiments (tools/ab_build.sh) point at another build of the
# same C ABI; the default is the in-tree library.
LIB_PATH = os.environ.get("PIXO_HIP_LIB") or os.path.join(_HERE, "libpixo_hip.so")


class JpegOptionsC(C.Structure):
    """pixo_jpeg_options (include/pixo_hip.h)"""
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("color_type", C.c_uint8), ("quality", C.c_uint8), ("subsampling", C.c_uint8),
        ("has_restart_interval", C.c_uint8), ("restart_interval", C.c_uint16),
        ("optimize_huffman", C.c_uint8), ("progressive", C.c_uint8), ("trellis_quant", C.c_uint8),
    ]


# every symbol include/pixo_hip.h declares
SYMBOLS = [
    "pixo_jpeg_options_from_preset", "pixo_hip_jpeg_encode", "pixo_hip_jpeg_encode_into",
    "pixo_hip_encode_jpeg", "pixo_hip_coeff_geometry", "pixo_hip_jpeg_coeffs",
    "pixo_hip_jpeg_coeffs_device", "pixo_hip_jpeg_coeffs_integer", "pixo_hip_jpeg_coeffs_integer_device", "pixo_hip_jpeg_entropy_encode", "pixo_hip_jpeg_entropy_encode_device",
    "pixo_hip_jpeg_encode_device", "pixo_hip_jpeg_encode_device_into", "pixo_hip_jpeg_encode_batch_device", "pixo_hip_jpeg_encode_batch_device_into", "pixo_hip_debug_lookback_fallbacks", "pixo_hip_debug_dispatch_gate", "pixo_hip_debug_stream_copy", "pixo_hip_debug_stream_io", "pixo_hip_debug_engine_clock", "pixo_hip_debug_scan_device_async", "pixo_hip_debug_scan_device_async_batch", "pixo_hip_png_filter", "pixo_hip_png_filter_device", "pixo_hip_png_filter_async",
    "pixo_hip_png_adler32_from_row_sums", "pixo_hip_band",
    "pixo_hip_band_encoder_create", "pixo_hip_band_encoder_destroy", "pixo_hip_band_encoder_rows",
    "pixo_hip_band_encoder_coeffs", "pixo_hip_band_encoder_count", "pixo_hip_band_encoder_lengths",
    "pixo_hip_band_encoder_pack", "pixo_hip_band_encoder_pack_device", "pixo_hip_band_encoder_copy_body",
    "pixo_hip_jpeg_splice", "pixo_hip_jpeg_splice_layout", "pixo_hip_jpeg_splice_finish", "pixo_hip_jpeg_band_count_host",
    "pixo_hip_jpeg_band_bits_host", "pixo_hip_jpeg_band_piece_host", "pixo_hip_jpeg_encode_multi", "pixo_hip_jpeg_encode_batch_multi",
    "pixo_hip_device_count", "pixo_hip_set_device", "pixo_hip_set_producer_stream", "pixo_hip_get_producer_stream", "pixo_hip_debug_configure", "pixo_hip_trim", "pixo_hip_free", "pixo_hip_copy_file",
    "pixo_hip_last_error", "pixo_hip_version",
]

_lib = None


def _preload_process_hip_runtime():
    """One HIP runtime per process.  libpixo_hip.so needs `libamdhip64.so.7`; PyTorch-ROCm
    wheels bundle their own copy under torch/lib with the same soname.  Whichever is mapped
    first serves both, and mapping ROCm's first makes a later `import torch` find no GPU.
    So when torch is installed, map ITS runtime before ours (without importing torch)."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass  # fall back to the loader's default resolution (RUNPATH -> /opt/rocm/lib)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "pixo_amd: %s is missing — build it with `make -C pixo_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback." % LIB_PATH)
    _preload_process_hip_runtime()
    L = C.CDLL(LIB_PATH)
    u8pp, szp = C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)
    optp = C.POINTER(JpegOptionsC)
    L.pixo_jpeg_options_from_preset.argtypes = [optp, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8]
    L.pixo_jpeg_options_from_preset.restype = None
    L.pixo_hip_jpeg_encode.argtypes = [C.c_void_p, C.c_size_t, optp, u8pp, szp]
    L.pixo_hip_jpeg_encode_into.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, optp, szp]
    L.pixo_hip_encode_jpeg.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint8,
                                       C.c_uint8, C.c_uint8, C.c_int, u8pp, szp]
    L.pixo_hip_coeff_geometry.argtypes = [C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8, szp, szp]
    L.pixo_hip_jpeg_coeffs.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8,
                                       C.c_uint8, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                       C.c_size_t]
    L.pixo_hip_jpeg_coeffs_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint8,
                                              C.c_uint8, C.c_uint8, C.c_uint32, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
    L.pixo_hip_jpeg_coeffs_integer.argtypes = L.pixo_hip_jpeg_coeffs.argtypes
    L.pixo_hip_jpeg_coeffs_integer_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint8,
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pixo_hip_jpeg_entropy_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, optp, u8pp, szp]
    L.pixo_hip_jpeg_entropy_encode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, optp, u8pp, szp]
    L.pixo_hip_jpeg_encode_device.argtypes = [C.c_void_p, optp, u8pp, szp]
    L.pixo_hip_jpeg_encode_device_into.argtypes = [C.c_void_p, optp, C.c_void_p, C.c_size_t, szp]
    L.pixo_hip_jpeg_encode_batch_device.argtypes = [C.c_void_p, optp, C.c_uint32, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.pixo_hip_jpeg_encode_batch_device_into.argtypes = [C.c_void_p, optp, C.c_uint32, C.c_void_p, C.c_size_t, szp, szp]
    L.pixo_hip_debug_lookback_fallbacks.restype = C.c_uint64
    if hasattr(L, "pixo_hip_debug_dispatch_gate"):  # (absent from A/B builds of older trees: tools/ab/)
        L.pixo_hip_debug_dispatch_gate.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pixo_hip_debug_stream_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    if hasattr(L, "pixo_hip_debug_engine_clock"):
        L.pixo_hip_debug_engine_clock.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    if hasattr(L, "pixo_hip_debug_stream_io"):
        L.pixo_hip_debug_stream_io.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.pixo_hip_debug_scan_device_async.argtypes = [C.c_void_p, optp, C.c_void_p, C.POINTER(C.c_int)]
    if hasattr(L, "pixo_hip_debug_scan_device_async_batch"):  # (absent from older builds dlopen'ed for A/B timing: PIXO_HIP_LIB)
        L.pixo_hip_debug_scan_device_async_batch.argtypes = [C.c_void_p, optp, C.c_uint32, C.c_void_p, C.POINTER(C.c_int)]
    L.pixo_hip_png_filter.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint32,
                                      C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    L.pixo_hip_png_filter_async.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pixo_hip_png_adler32_from_row_sums.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    L.pixo_hip_png_adler32_from_row_sums.restype = C.c_uint32
    L.pixo_hip_png_filter_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint32,
                                             C.c_void_p, C.POINTER(C.c_uint32)]
    L.pixo_hip_band.argtypes = [C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint32, C.c_uint32,
                                C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), szp, szp, szp, szp]
    i16p, u64p = C.POINTER(C.c_int16), C.POINTER(C.c_uint64)
    L.pixo_hip_band_encoder_create.argtypes = [optp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
    L.pixo_hip_band_encoder_destroy.argtypes = [C.c_void_p]
    L.pixo_hip_band_encoder_destroy.restype = None
    L.pixo_hip_band_encoder_rows.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.pixo_hip_band_encoder_coeffs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, i16p]
    L.pixo_hip_band_encoder_count.argtypes = [C.c_void_p, i16p, u64p]
    L.pixo_hip_band_encoder_lengths.argtypes = [C.c_void_p, i16p, u64p, u64p]
    L.pixo_hip_band_encoder_pack.argtypes = [C.c_void_p, C.c_uint64, u8pp, szp]
    L.pixo_hip_band_encoder_pack_device.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_void_p), szp]
    L.pixo_hip_band_encoder_copy_body.argtypes = [C.c_void_p, C.c_void_p]
    L.pixo_hip_jpeg_splice_layout.argtypes = [optp, u64p, C.c_void_p, C.c_uint32, szp, szp]
    L.pixo_hip_jpeg_splice_finish.argtypes = [optp, u64p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
    L.pixo_hip_jpeg_splice.argtypes = [optp, u64p, C.POINTER(C.c_void_p), szp, C.c_uint32, u8pp, szp]
    L.pixo_hip_jpeg_band_count_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, optp, C.c_uint32, i16p, u64p]
    L.pixo_hip_jpeg_band_bits_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, optp, C.c_uint32, i16p, u64p, u64p]
    L.pixo_hip_jpeg_band_piece_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, optp, C.c_uint32, i16p, u64p, C.c_uint64,
                                                u8pp, szp]
    L.pixo_hip_jpeg_encode_multi.argtypes = [C.c_void_p, C.c_size_t, optp, C.POINTER(C.c_int), C.c_uint32, u8pp, szp]
    L.pixo_hip_jpeg_encode_batch_multi.argtypes = [C.c_void_p, optp, C.c_uint32, C.POINTER(C.c_int), C.c_uint32, C.c_void_p, C.c_size_t, szp, szp]
    L.pixo_hip_set_producer_stream.argtypes = [C.c_void_p]
    L.pixo_hip_get_producer_stream.restype = C.c_void_p
    L.pixo_hip_debug_configure.argtypes = [C.c_char_p]
    L.pixo_hip_device_count.restype = C.c_int
    L.pixo_hip_set_device.argtypes = [C.c_int]
    L.pixo_hip_free.argtypes = [C.c_void_p]
    L.pixo_hip_free.restype = None
    L.pixo_hip_copy_file.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.pixo_hip_copy_file.restype = None
    L.pixo_hip_last_error.restype = C.c_char_p
    L.pixo_hip_version.restype = C.c_char_p
    for name in ("pixo_hip_jpeg_encode", "pixo_hip_jpeg_encode_into", "pixo_hip_encode_jpeg",
                 "pixo_hip_coeff_geometry", "pixo_hip_jpeg_coeffs", "pixo_hip_jpeg_coeffs_device",
                 "pixo_hip_jpeg_entropy_encode", "pixo_hip_band", "pixo_hip_set_device"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


_bytes_new = C.pythonapi.PyBytes_FromStringAndSize
_bytes_new.restype = C.py_object
_bytes_new.argtypes = [C.c_void_p, C.c_ssize_t]


def file_bytes(L, ptr, n: int) -> bytes:
    """The n bytes at ptr as a Python bytes object.  Large files are copied by the library's copy threads into a bytes
    object created uninitialised (ours alone until it is returned), with a huge-page hint before its first touch:
    ctypes.string_at on a 178 MB file is a one-thread memcpy into 43,000 fresh pages, 28 ms of a 45 ms call."""
    if n < (2 << 20):
        return C.string_at(ptr, n)
    b = _bytes_new(None, n)
    L.pixo_hip_copy_file(C.cast(C.c_char_p(b), C.c_void_p), ptr, n)
    return b
