"""ctypes binding for tests/emu/libpixo_emu.so: the DEVICE tile code (pixo_amd/csrc/
jpeg_tile.h) compiled for the host and driven lane by lane.  Test harness only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_EMU = os.path.join(_HERE, "emu")
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-C", _EMU], stdout=subprocess.DEVNULL)
        L = C.CDLL(os.path.join(_EMU, "libpixo_emu.so"))
        L.emu_jpeg_coeffs.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.emu_quant_mismatches.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int]
        L.emu_quant_mismatches.restype = C.c_long
        L.emu_quant_fastpath_audit.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def geometry(w, h, color_type, subsampling):
    gray = color_type == 0
    s420 = (not gray) and subsampling == 1
    unit = 16 if s420 else 8
    ux, uy = (w + unit - 1) // unit, (h + unit - 1) // unit
    units = ux * uy
    return (4 * units if s420 else units), (0 if gray else units)


def coeffs(pixels, w, h, color_type=2, subsampling=1, quality=80, allow_fast=True, misalign=0, wave_order=0):
    px = np.ascontiguousarray(pixels, np.uint8)
    if misalign:
        buf = np.empty(px.size + 16, np.uint8)
        off = (-buf.ctypes.data) % 4 + misalign
        buf[off:off + px.size] = px
        px = buf[off:off + px.size]
    yb, cbn = geometry(w, h, color_type, subsampling)
    y = np.full((yb, 64), -32768, np.int16)
    cb = np.full((max(cbn, 1), 64), -32768, np.int16)
    cr = np.full((max(cbn, 1), 64), -32768, np.int16)
    stats = (C.c_long * 3)()
    rc = lib().emu_jpeg_coeffs(px.ctypes.data, w, h, color_type, subsampling, quality, y.ctypes.data,
                               cb.ctypes.data, cr.ctypes.data, int(allow_fast), stats, wave_order)
    assert rc == 0, "the emulated kernel issued %d vector load(s) outside the image's bytes" % -rc  # (jpeg_tile.h emu_check_load)
    if wave_order == 0:
        # the device has the DCT passes and the quantiser in two forms (packed: launches of several generations; scalar: one
        # generation — jpeg_tile.h block_rows): wave order 0 ran the packed one, order 1 runs the scalar one: same tuple
        y2, cb2, cr2, _ = coeffs(pixels, w, h, color_type, subsampling, quality, allow_fast, misalign, wave_order=1)
        assert np.array_equal(y, y2) and np.array_equal(cb[:cbn], cb2) and np.array_equal(cr[:cbn], cr2), "scalar and packed forms differ"
    return y, cb[:cbn], cr[:cbn], (stats[0], stats[1], stats[2])
