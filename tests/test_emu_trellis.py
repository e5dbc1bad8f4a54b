"""The DEVICE trellis quantiser (pixo_amd/csrc/jpeg_trellis.h) compiled for the host against the
oracle's restatement of src/jpeg/trellis.rs (itself pinned on reference-made preset-2 files):
identical i16 blocks on random and on structured DCT input."""
import ctypes as C

import numpy as np
import pytest

import emu_lib as E
import oracle_lib as O


def _both(dct, q):
    n = dct.shape[0]
    L = E.lib()
    got = np.zeros((n, 64), np.int16)
    fast = np.ones((n, 64), np.int16)
    lanes = np.full((n, 64), 2, np.int16)
    for fn, dst in ((L.emu_trellis, got), (L.emu_trellis_fast, fast), (L.emu_trellis_lanes, lanes)):
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        fn.restype = None
        fn(dct.ctypes.data, q.ctypes.data, n, dst.ctypes.data)
    assert np.array_equal(got, fast)  # reference-shaped search == register-resident restatement
    assert np.array_equal(got, lanes)  # ... == the eight-lane form's per-lane functions (round 5), exchanges as loops
    OL = O.lib()
    OL.po_trellis_quantize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    OL.po_trellis_quantize.restype = None
    want = np.zeros((n, 64), np.int16)
    for b in range(n):
        OL.po_trellis_quantize(dct[b].ctypes.data, q.ctypes.data, want[b].ctypes.data)
    return got, want


@pytest.mark.parametrize("scale", [0.5, 3.0, 20.0, 150.0, 900.0])
@pytest.mark.parametrize("qkind", ["flat1", "std80", "coarse"])
def test_device_trellis_equals_oracle(scale, qkind):
    rng = np.random.RandomState(int(scale * 10) + len(qkind))
    dct = (rng.standard_normal((300, 64)) * scale).astype(np.float32)
    dct[::7] = np.round(dct[::7])            # exact integers and ties
    dct[::11, 20:] = 0                        # long zero runs
    dct[::13] = (np.round(dct[::13] * 2) / 2).astype(np.float32)  # exact halves: floor/round/ceil ties
    q = {"flat1": np.ones(64, np.float32), "std80": (6 + np.arange(64) // 2).astype(np.float32),
         "coarse": np.full(64, 40, np.float32)}[qkind]
    got, want = _both(np.ascontiguousarray(dct), q)
    assert np.array_equal(got, want)


def test_sparse_blocks_with_sixteen_zero_runs():
    dct = np.zeros((40, 64), np.float32)
    rng = np.random.RandomState(5)
    zz = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
    for b in range(40):
        for k in rng.choice(np.arange(1, 64), size=1 + b % 4, replace=False):
            dct[b, zz[k]] = rng.uniform(-60, 60)
        dct[b, 0] = rng.uniform(-500, 500)
    q = np.full(64, 8, np.float32)
    got, want = _both(dct, q)
    assert np.array_equal(got, want)
