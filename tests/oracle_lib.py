"""ctypes binding for oracle/libpixo_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (pixo_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

GRAY, RGB = 0, 2
S444, S420 = 0, 1


class Options(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("color_type", C.c_uint8), ("quality", C.c_uint8), ("subsampling", C.c_uint8),
        ("has_restart", C.c_uint8), ("restart_interval", C.c_uint16),
        ("optimize_huffman", C.c_uint8), ("progressive", C.c_uint8), ("trellis_quant", C.c_uint8),
    ]


def build():
    so = os.path.join(ORACLE_DIR, "libpixo_oracle.so")
    src = os.path.join(ORACLE_DIR, "pixo_oracle.c")
    if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libpixo_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u8p, i16p = C.POINTER(C.c_uint8), C.POINTER(C.c_int16)
        L.po_jpeg_coeffs.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8,
                                     C.c_uint8, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.po_jpeg_coeffs.restype = C.c_int
        L.po_coeff_geometry.argtypes = [C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8,
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.po_encode_jpeg.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Options),
                                     C.POINTER(u8p), C.POINTER(C.c_size_t)]
        L.po_encode_jpeg.restype = C.c_int
        L.po_encode_jpeg_from_coeffs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.POINTER(Options), C.POINTER(u8p),
                                                 C.POINTER(C.c_size_t)]
        L.po_encode_jpeg_from_coeffs.restype = C.c_int
        L.po_encode_jpeg_flat.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32,
                                          C.c_uint8, C.c_uint8, C.c_uint8, C.c_int,
                                          C.POINTER(u8p), C.POINTER(C.c_size_t)]
        L.po_encode_jpeg_flat.restype = C.c_int
        L.po_symbol_histograms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.POINTER(Options), C.c_void_p, C.c_void_p]
        L.po_free.argtypes = [C.c_void_p]
        L.po_strerror.restype = C.c_char_p
        L.po_dct_2d.argtypes = [C.c_void_p, C.c_void_p]
        L.po_quantize_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.po_quant_tables.argtypes = [C.c_uint8, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.po_rgb_to_ycbcr.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, C.c_void_p]
        _LIB = L
    return _LIB


class OracleError(Exception):
    def __init__(self, code):
        self.code = code
        super().__init__(lib().po_strerror(code).decode())


def geometry(w, h, color_type, subsampling):
    yb, cb = C.c_size_t(), C.c_size_t()
    lib().po_coeff_geometry(w, h, color_type, subsampling, C.byref(yb), C.byref(cb))
    return yb.value, cb.value


def coeffs(pixels, w, h, color_type=RGB, subsampling=S420, quality=80, threads=1):
    """-> (y[yb,64], cb[cbn,64], cr[cbn,64]) int16, natural order (a15 tuple)."""
    px = np.ascontiguousarray(pixels, dtype=np.uint8)
    yb, cbn = geometry(w, h, color_type, subsampling)
    y = np.empty((yb, 64), np.int16)
    cb = np.empty((cbn, 64), np.int16)
    cr = np.empty((cbn, 64), np.int16)
    rc = lib().po_jpeg_coeffs(px.ctypes.data, w, h, color_type, subsampling, quality,
                              y.ctypes.data, cb.ctypes.data, cr.ctypes.data, threads)
    if rc:
        raise OracleError(rc)
    return y, cb, cr


def coeffs_integer(pixels, w, h, color_type=RGB, quality=80):
    """The integer DCT family composed per 8x8 block (oracle/pixo_int_oracle.c): 4:4:4 / gray tuple."""
    px = np.ascontiguousarray(pixels, dtype=np.uint8)
    yb, cbn = geometry(w, h, color_type, S444)
    y = np.empty((yb, 64), np.int16)
    cb = np.empty((cbn, 64), np.int16)
    cr = np.empty((cbn, 64), np.int16)
    L = lib()
    L.po_jpeg_coeffs_integer.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.po_jpeg_coeffs_integer(px.ctypes.data, w, h, color_type, quality, y.ctypes.data, cb.ctypes.data, cr.ctypes.data)
    if rc:
        raise OracleError(rc)
    return y, cb, cr


def make_options(w, h, color_type=RGB, quality=80, subsampling=S420, restart=None,
                 optimize_huffman=False, progressive=False, trellis=False):
    o = Options()
    o.width, o.height, o.color_type, o.quality, o.subsampling = w, h, color_type, quality, subsampling
    o.has_restart = 0 if restart is None else 1
    o.restart_interval = 0 if restart is None else restart
    o.optimize_huffman, o.progressive, o.trellis_quant = int(optimize_huffman), int(progressive), int(trellis)
    return o


def _take(outp, n):
    data = C.string_at(outp, n.value)
    lib().po_free(outp)
    return data


def encode(pixels, opts):
    px = np.ascontiguousarray(pixels, dtype=np.uint8)
    outp, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    rc = lib().po_encode_jpeg(px.ctypes.data, px.size, C.byref(opts), C.byref(outp), C.byref(n))
    if rc:
        raise OracleError(rc)
    return _take(outp, n)


def encode_from_coeffs(y, cb, cr, opts):
    y = np.ascontiguousarray(y, np.int16)
    cb = np.ascontiguousarray(cb, np.int16)
    cr = np.ascontiguousarray(cr, np.int16)
    outp, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    rc = lib().po_encode_jpeg_from_coeffs(y.ctypes.data, cb.ctypes.data, cr.ctypes.data,
                                          C.byref(opts), C.byref(outp), C.byref(n))
    if rc:
        raise OracleError(rc)
    return _take(outp, n)


def encode_flat(pixels, w, h, color_type, quality, preset, s420):
    px = np.ascontiguousarray(pixels, dtype=np.uint8)
    outp, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    rc = lib().po_encode_jpeg_flat(px.ctypes.data, px.size, w, h, color_type, quality, preset,
                                   int(bool(s420)), C.byref(outp), C.byref(n))
    if rc:
        raise OracleError(rc)
    return _take(outp, n)


def histograms(y, cb, cr, opts):
    dc = np.zeros((2, 12), np.uint64)
    ac = np.zeros((2, 256), np.uint64)
    lib().po_symbol_histograms(np.ascontiguousarray(y).ctypes.data,
                               np.ascontiguousarray(cb).ctypes.data,
                               np.ascontiguousarray(cr).ctypes.data, C.byref(opts),
                               dc.ctypes.data, ac.ctypes.data)
    return dc, ac


# ---- PNG row filters + Adler-32 (oracle/pixo_png_oracle.c) ---------------------------------------
S_NONE, S_SUB, S_UP, S_AVERAGE, S_PAETH, S_MINSUM, S_ADAPTIVE, S_ADAPTIVE_FAST, S_BIGRAMS = range(9)


def png_filter(pixels, w, h, bpp, strategy, stateful_fast=False):
    """-> (filtered stream bytes [h * (w*bpp + 1)], adler32)"""
    L = lib()
    L.po_png_filter.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.po_png_filter.restype = C.c_int
    px = np.ascontiguousarray(pixels, dtype=np.uint8)
    assert px.size == w * h * bpp
    out = np.empty(h * (w * bpp + 1), np.uint8)
    ad = C.c_uint32()
    rc = L.po_png_filter(px.ctypes.data, w, h, bpp, strategy, int(stateful_fast), out.ctypes.data, C.byref(ad))
    if rc:
        raise OracleError(rc)
    return out, ad.value
