"""400 randomised cases whose answers were made by the REFERENCE's wasm build (tools/oracle_vs_wasm.py --record, which also
compared 120,000 such cases with the oracle in the build container: 0 mismatches): lengths + sha256 in
tests/golden/jpeg_fresh_cases.json, inputs regenerated from the case number (tests/fresh_cases.py).
CPU: the oracle's restatement must give the reference's files.  GPU (-m gpu): so must the HIP library through the C ABI."""
import hashlib
import json
import os

import pytest

import fresh_cases as F
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "jpeg_fresh_cases.json")))["cases"]
CHUNKS = [CASES[i:i + 50] for i in range(0, len(CASES), 50)]


def _same_options(c, o):
    return all(c[k] == o[k] for k in ("kind", "w", "h", "color_type", "quality", "preset", "s420"))


def _check(c, blob):
    assert len(blob) == c["len"], c
    assert hashlib.sha256(blob).hexdigest() == c["sha256"], c


def test_generator_reproduces_the_recorded_cases():
    kinds, presets = set(), set()
    for c in CASES:
        o, px = F.case_of(c["id"])
        assert _same_options(c, o), (c, o)
        kinds.add(c["kind"]); presets.add(c["preset"])
    assert kinds == set(F.KINDS) and presets == {0, 1, 2}


@pytest.mark.parametrize("chunk", CHUNKS, ids=["%d-%d" % (ch[0]["id"], ch[-1]["id"]) for ch in CHUNKS])
def test_oracle_gives_the_reference_files(chunk):
    for c in chunk:
        o, px = F.case_of(c["id"])
        _check(c, bytes(O.encode_flat(px, c["w"], c["h"], c["color_type"], c["quality"], c["preset"], c["s420"])))


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", CHUNKS, ids=["%d-%d" % (ch[0]["id"], ch[-1]["id"]) for ch in CHUNKS])
def test_hip_library_gives_the_reference_files(chunk):
    from pixo_amd import jpeg
    for c in chunk:
        o, px = F.case_of(c["id"])
        _check(c, bytes(jpeg.encode_jpeg(px, c["w"], c["h"], c["color_type"], c["quality"], c["preset"], c["s420"])))


# ---- PNG row filters (C5): 300 cases recorded by tools/oracle_vs_wasm_png.py (99,770 compared there, 0 mismatches) ----
PNG_CASES = json.load(open(os.path.join(HERE, "golden", "png_fresh_cases.json")))["cases"]
PNG_CHUNKS = [PNG_CASES[i:i + 50] for i in range(0, len(PNG_CASES), 50)]


def _png_check(c, flt, adler):
    assert flt.size == c["filtered_len"], c
    assert hashlib.sha256(flt.tobytes()).hexdigest() == c["filtered_sha256"], c
    assert adler == c["adler32"], c


def _png_input(c):
    o, px = F.png_case_of(c["id"])
    assert all(c[k] == o[k] for k in ("kind", "w", "h", "color_type", "preset")), (c, o)
    return px


@pytest.mark.parametrize("chunk", PNG_CHUNKS, ids=["%d-%d" % (ch[0]["id"], ch[-1]["id"]) for ch in PNG_CHUNKS])
def test_png_oracle_gives_the_reference_streams(chunk):
    for c in chunk:
        strategy, stateful = {0: (O.S_ADAPTIVE_FAST, True), 1: (O.S_ADAPTIVE, False), 2: (O.S_BIGRAMS, False)}[c["preset"]]
        _png_check(c, *O.png_filter(_png_input(c), c["w"], c["h"], F.PNG_BPP[c["color_type"]], strategy, stateful))


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", PNG_CHUNKS, ids=["%d-%d" % (ch[0]["id"], ch[-1]["id"]) for ch in PNG_CHUNKS])
def test_png_hip_library_gives_the_reference_streams(chunk):
    from pixo_amd import png
    for c in chunk:
        strategy, flags = {0: (png.FilterStrategy.ADAPTIVE_FAST, png.NO_RAYON), 1: (png.FilterStrategy.ADAPTIVE, 0),
                           2: (png.FilterStrategy.BIGRAMS, 0)}[c["preset"]]
        _png_check(c, *png.apply_filters(_png_input(c), c["w"], c["h"], F.PNG_BPP[c["color_type"]], strategy, flags))
