"""The oracle's restatement of the progressive + trellis path (preset 2) against files made by the
reference's own wasm build (tests/golden/make_golden_p2.py): byte-identical."""
import hashlib
import json
import os
import sys

import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG  # noqa: E402

CASES = json.load(open(os.path.join(HERE, "golden", "jpeg_p2_cases.json")))["cases"]


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_oracle_preset2_files(c):
    px = MG.GEN[c["gen"]](c["w"], c["h"], c["seed"])
    got = O.encode_flat(px, c["w"], c["h"], c["color_type"], c["quality"], 2, c["s420"])
    assert len(got) == c["len"]
    assert hashlib.sha256(got).hexdigest() == c["sha256"]
    if c.get("stored"):
        assert got == open(os.path.join(HERE, "golden", "jpeg_p2", c["name"] + ".jpg"), "rb").read()
