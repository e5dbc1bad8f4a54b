"""400 randomised cases whose answers were made by the REFERENCE's wasm build (tools/oracle_vs_wasm.py --record, which also
compared 20,000 such cases with the oracle in the build container: 0 mismatches): lengths + sha256 in
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
