"""The PNG filter kernel's per-group arithmetic (pixo_amd/csrc/png_filter_math.h: neighbour alignment for every
pixel size, SWAR subtract / average, packed 16-bit Paeth, scores, the reference's decision sequences, checksum
terms) compiled for the host and driven row by row against the oracle — no GPU needed.  The kernel's loads,
reductions and write-out are covered by tests/test_gpu_png.py."""
import ctypes as C
import zlib

import numpy as np
import pytest

import emu_lib as E
import oracle_lib as O
import synth


def _emu(px, w, h, bpp, strategy):
    L = E.lib()
    L.emu_png_filter.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_void_p]
    L.emu_png_filter.restype = C.c_long
    out = np.zeros(h * (w * bpp + 1), np.uint8)
    ad = L.emu_png_filter(px.ctypes.data, w, h, bpp, strategy, out.ctypes.data)
    return out, ad & 0xFFFFFFFF


@pytest.mark.parametrize("bpp", [1, 2, 3, 4, 6, 8])
@pytest.mark.parametrize("strategy", [0, 1, 2, 3, 4, 5, 6, 7, 8])
def test_group_arithmetic_against_the_oracle(bpp, strategy):
    for (w, h, seed) in [(67, 41, 1), (256, 40, 2), (333, 35, 3), (5, 200, 4), (1, 70, 5)]:
        px = synth.lcg_bytes(w * h * bpp, seed + bpp)
        if seed % 2 == 0:  # smoother content: other filters win
            px = (np.cumsum(px.astype(np.int64) % 5) % 256).astype(np.uint8)
        if w * h <= 4096 and strategy in (6, 7, 8):
            continue  # the launcher turns these into Sub (host logic, tested on the C ABI)
        want, wad = O.png_filter(px, w, h, bpp, strategy, stateful_fast=False)
        got, gad = _emu(px, w, h, bpp, strategy)
        assert np.array_equal(got, want), (w, h, seed)
        assert gad == wad == zlib.adler32(want.tobytes())


def test_paeth_on_every_triple_and_swar_on_every_pair():
    """All 2^24 (left, up, up-left) triples through the packed 16-bit Paeth predictor, and with them every
    (sample, predictor) byte pair through the SWAR subtract / average."""
    pairs, per = 1024, 16384  # same layout as the GPU test, 1024 row pairs of 16384 triples
    w, h = 3 * per, 2 * pairs
    rng = np.random.RandomState(12)
    img = rng.randint(0, 256, (h, w)).astype(np.uint8)
    t = (np.arange(pairs, dtype=np.uint32)[:, None] * per + np.arange(per, dtype=np.uint32)[None, :])
    img[0::2, 0::3] = (t >> 16).astype(np.uint8)
    img[0::2, 1::3] = ((t >> 8) & 255).astype(np.uint8)
    img[1::2, 0::3] = (t & 255).astype(np.uint8)
    px = img.reshape(-1)
    for strategy in (O.S_PAETH, O.S_AVERAGE, O.S_SUB):
        want, wad = O.png_filter(px, w, h, 1, strategy)
        got, gad = _emu(px, w, h, 1, strategy)
        assert np.array_equal(got, want) and gad == wad
