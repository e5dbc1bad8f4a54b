"""The INTEGER secondary mode: the reference's fixed-point DCT family, which `encode()` never calls.

No runnable reference exists for it (dead code is stripped from the wasm build), so the oracle's restatement
(oracle/pixo_int_oracle.c) is PINNED ONLY on the exact values the reference's own unit tests state
(src/jpeg/dct.rs:867-1183, src/simd/x86_64.rs:2077-2250) — transliterated below — and on the properties they assert.
The device arithmetic (pixo_amd/csrc/jpeg_int_math.h, compiled for the host) must equal the oracle bit for bit; the GPU
kernel itself is compared with the oracle in tests/test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

import emu_lib as E
import oracle_lib as O


def _oracle():
    L = O.lib()
    L.po_dct_2d_integer.argtypes = L.po_dct_2d_fast.argtypes = [C.c_void_p, C.c_void_p]
    L.po_quantize_block_integer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.po_rgb_to_ycbcr_2p16.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, C.c_void_p]
    return L


def dct_integer(block, fast=False, emu=False):
    b = np.ascontiguousarray(block, np.int16)
    out = np.zeros(64, np.int32)
    if emu:
        L = E.lib()
        L.emu_int_dct_fast.argtypes = [C.c_void_p, C.c_void_p]
        L.emu_int_dct_fast(b.ctypes.data, out.ctypes.data)
    else:
        (_oracle().po_dct_2d_fast if fast else _oracle().po_dct_2d_integer)(b.ctypes.data, out.ctypes.data)
    return out


def quantize(dct, q, emu=False):
    d = np.ascontiguousarray(dct, np.int32)
    t = np.ascontiguousarray(q, np.uint16)
    out = np.zeros(64, np.int16)
    if emu:
        L = E.lib()
        L.emu_int_quantize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.emu_int_quantize(d.ctypes.data, t.ctypes.data, out.ctypes.data)
    else:
        _oracle().po_quantize_block_integer(d.ctypes.data, t.ctypes.data, out.ctypes.data)
    return out


def color(r, g, b, emu=False):
    out = np.zeros(3, np.int32)
    if emu:
        L = E.lib()
        L.emu_int_color.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.emu_int_color(r, g, b, out.ctypes.data)
    else:
        _oracle().po_rgb_to_ycbcr_2p16(r, g, b, out.ctypes.data)
    return out


# ---- the reference's unit tests, transliterated (exact values where it states them) --------------------------------
def test_integer_dct_zeros_constant_and_extremes():
    assert not dct_integer(np.zeros(64)).any()                                   # dct.rs:874-881
    r = dct_integer(np.full(64, 100))                                            # :884-897
    assert r[0] > 100 and (np.abs(r[1:]) <= 1).all()
    assert dct_integer(np.full(64, 127))[0] > 0 and dct_integer(np.full(64, -128))[0] < 0   # :1168-1183


def test_integer_dct_gradient_stripes_and_checkerboard_properties():
    rows, cols = np.indices((8, 8))
    grad = np.clip((rows + cols) * 16 - 112, -128, 127).reshape(-1)             # :900-933
    r = dct_integer(grad).astype(np.int64)
    assert abs(r[0]) < 50 and (r[:16] ** 2).sum() > (r[48:] ** 2).sum()
    checker = np.where((rows + cols) % 2 == 0, 100, -100).reshape(-1)            # :1113-1131
    assert (dct_integer(checker)[1:].astype(np.int64) ** 2).sum() > 0
    assert abs(dct_integer(np.where(rows % 2 == 0, 100, -100).reshape(-1))[0]) < 10   # :1134-1149
    assert abs(dct_integer(np.where(cols % 2 == 0, 100, -100).reshape(-1))[0]) < 10   # :1152-1166


def test_dct_2d_fast_constant_block_shortcut_and_agreement_with_the_integer_transform():
    for v in (0, 50, -30, 127, -128):                                            # :991-1049: DC = 8 * value, AC = 0
        r = dct_integer(np.full(64, v), fast=True)
        assert r[0] == 8 * v and not r[1:].any()
        assert np.array_equal(r, dct_integer(np.full(64, v)))                    # the shortcut is what the transform gives anyway
    pattern = (np.arange(64) * 7) % 256 - 128                                    # :965-988: fast == integer off aarch64
    assert np.array_equal(dct_integer(pattern, fast=True), dct_integer(pattern))
    ramp = np.arange(64) - 32                                                    # :1052-1070
    assert np.count_nonzero(dct_integer(ramp, fast=True)[1:]) > 0


def test_quantize_block_integer_exact_values():
    d = np.zeros(64, np.int32); d[:4] = [100, -50, 75, -25]                      # :1073-1092
    assert list(quantize(d, np.full(64, 16))[:4]) == [6, -3, 5, -2]
    d = np.full(64, 100, np.int32)                                               # :1095-1110
    assert quantize(d, np.full(64, 8))[0] == 13 and quantize(d, np.full(64, 64))[0] == 2
    d = np.zeros(64, np.int32); d[:3] = [16, 8, 7]                               # edge values: exactly half rounds away from zero
    assert list(quantize(d, np.full(64, 16))[:3]) == [1, 1, 0]
    one = np.zeros(64, np.int16); one[0] = 100                                   # :936-952
    assert quantize(dct_integer(one), np.full(64, 16))[0] != 0


def test_rgb_to_ycbcr_2p16_black_white_red():
    assert list(color(0, 0, 0)) == [-128, 0, 0]                                  # x86_64.rs:2158-2176 (|y + 128| < 1, |cb|, |cr| < 1)
    y, cb, cr = color(255, 255, 255)                                             # :2179-2197
    assert abs(y - 127) < 1 and abs(cb) < 1 and abs(cr) < 1
    y, cb, cr = color(255, 0, 0)                                                 # :2200-2231
    assert -100 < y < 0 and cb < 0 and cr > 0
    for i in range(10):                                                          # :2234-2250
        px = [(3 * i + k) * 7 % 256 for k in range(3)]
        assert all(abs(int(v)) < 200 for v in color(*px))


# ---- the device arithmetic (host-compiled) against the oracle -------------------------------------------------------
def test_device_integer_math_equals_the_oracle():
    rng = np.random.RandomState(7)
    blocks = [rng.randint(-128, 128, 64) for _ in range(300)] + [np.full(64, v) for v in (-128, -1, 0, 1, 127)]
    blocks += [np.where(np.arange(64) % 2 == 0, 127, -128), np.arange(64) * 4 - 128]
    for b in blocks:
        want = dct_integer(b, fast=True)
        assert np.array_equal(dct_integer(b, emu=True), want)
        for q in (1, 2, 7, 16, 99, 255):
            assert np.array_equal(quantize(want, np.full(64, q), emu=True), quantize(want, np.full(64, q)))
    for rgb in rng.randint(0, 256, (2000, 3)).tolist() + [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255]]:
        assert np.array_equal(color(*rgb, emu=True), color(*rgb))


def test_integer_tables_are_the_float_tables():
    L = _oracle()
    L.po_quant_tables_int.argtypes = [C.c_uint8, C.c_void_p, C.c_void_p]
    for q in (1, 50, 80, 100):
        lum, chr_ = np.zeros(64, np.uint16), np.zeros(64, np.uint16)
        L.po_quant_tables_int(q, lum.ctypes.data, chr_.ctypes.data)
        assert lum.min() >= 1 and lum.max() <= 255 and chr_.min() >= 1
    lum = np.zeros(64, np.uint16); chr_ = np.zeros(64, np.uint16)
    L.po_quant_tables_int(50, lum.ctypes.data, chr_.ctypes.data)
    assert lum[0] == 16  # quantize.rs:287-291: q50 -> lum[0] = 16
