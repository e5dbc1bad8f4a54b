"""oracle/pixo_png_oracle.c (row filters, strategy rules, Adler-32) against vectors made by the
REFERENCE's own wasm build (tests/golden/make_golden_png.py): filter byte of every row, the
filtered stream (sha256, small ones verbatim) and the zlib trailer.  Pins the PNG oracle."""
import hashlib
import json
import os
import sys
import zlib

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_png as MG  # noqa: E402  (input generators only; nothing is run)

CASES = json.load(open(os.path.join(HERE, "golden", "png_cases.json")))["cases"]
BPP = {0: 1, 1: 2, 2: 3, 3: 4}


def run_case(c, fn):
    px = MG.make_input(c)
    # preset 0 = AdaptiveFast as a build without rayon runs it (stateful); 1 = Adaptive; 2 = Bigrams
    strategy, stateful = {0: (O.S_ADAPTIVE_FAST, True), 1: (O.S_ADAPTIVE, False), 2: (O.S_BIGRAMS, False)}[c["preset"]]
    flt, adler = fn(px, c["w"], c["h"], BPP[c["color_type"]], strategy, stateful)
    row = c["w"] * BPP[c["color_type"]] + 1
    assert "".join(str(int(f)) for f in flt[::row]) == c["filters"]
    assert flt.size == c["filtered_len"]
    assert hashlib.sha256(flt.tobytes()).hexdigest() == c["filtered_sha256"]
    assert adler == c["adler32"] == zlib.adler32(flt.tobytes())
    if c.get("stored"):
        assert flt.tobytes() == open(os.path.join(HERE, "golden", "png", c["name"] + ".flt"), "rb").read()


SMALL = [c for c in CASES if c["w"] * c["h"] <= 2100 * 1100]
BIG = [c for c in CASES if c["w"] * c["h"] > 2100 * 1100]


@pytest.mark.parametrize("c", SMALL, ids=[c["name"] for c in SMALL])
def test_png_oracle_reproduces_reference_filtered_stream(c):
    run_case(c, O.png_filter)


@pytest.mark.parametrize("c", BIG, ids=[c["name"] for c in BIG])
def test_png_oracle_config5_4096_rgba(c):
    # filtered stream sha256 240e005d..., Adler-32 0x90cc12e3, {None 1316, Sub 800, Up 436, Avg 1048, Paeth 496}
    assert c["adler32"] == 0x90CC12E3 and c["filtered_sha256"].startswith("240e005d4da54561")
    run_case(c, O.png_filter)


def test_fixed_strategies_and_adler_against_zlib_and_definitions():
    rng = np.random.RandomState(1)
    w, h, bpp = 37, 11, 3
    px = rng.randint(0, 256, w * h * bpp).astype(np.uint8)
    img = px.reshape(h, w * bpp).astype(np.int32)
    prev = np.vstack([np.zeros((1, w * bpp), np.int32), img[:-1]])
    left = np.hstack([np.zeros((h, bpp), np.int32), img[:, :-bpp]])
    ul = np.hstack([np.zeros((h, bpp), np.int32), prev[:, :-bpp]])
    want = {O.S_NONE: img, O.S_SUB: img - left, O.S_UP: img - prev, O.S_AVERAGE: img - (left + prev) // 2}
    p = left + prev - ul
    pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
    want[O.S_PAETH] = img - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
    for s, arr in want.items():
        flt, adler = O.png_filter(px, w, h, bpp, s)
        rows = flt.reshape(h, w * bpp + 1)
        assert (rows[:, 0] == s).all() and np.array_equal(rows[:, 1:], (arr & 0xFF).astype(np.uint8))
        assert adler == zlib.adler32(flt.tobytes())
