"""Pins the CPU oracle (oracle/pixo_oracle.c) to the REFERENCE: every golden
vector made by the reference's own wasm build must be reproduced byte-for-byte.
CPU-only."""
import pytest

import golden_util as G
import oracle_lib as O

SMALL = G.cases(max_pixels=1100 * 1100)
LARGE = G.cases(min_pixels=1100 * 1100 + 1, max_pixels=4096 * 4096)


@pytest.mark.parametrize("c", SMALL, ids=[c["name"] for c in SMALL])
def test_oracle_matches_reference_small(c):
    blob = O.encode_flat(G.make_input(c), c["w"], c["h"], c["color_type"], c["quality"],
                         c["preset"], c["s420"])
    G.check(c, blob)


@pytest.mark.slow
@pytest.mark.parametrize("c", LARGE, ids=[c["name"] for c in LARGE])
def test_oracle_matches_reference_large(c):
    blob = O.encode_flat(G.make_input(c), c["w"], c["h"], c["color_type"], c["quality"],
                         c["preset"], c["s420"])
    G.check(c, blob)


def test_oracle_error_order_matches_reference():
    """Validation order of jpeg/mod.rs:333-373 (messages recorded from the wasm)."""
    import synth
    want = {"Invalid quality": O.lib().po_strerror(-1), }
    for e in G.load()["errors"]:
        data = synth.lcg_bytes(e["nbytes"], 3)
        with pytest.raises(O.OracleError) as ei:
            O.encode_flat(data, e["w"], e["h"], e["color_type"], e["quality"], e["preset"], e["s420"])
        msg = e["error"]
        code = ei.value.code
        if msg.startswith("Invalid quality"):
            assert code == -1
        elif msg.startswith("Invalid image dimensions"):
            assert code == -3
        elif msg.startswith("Invalid color type") or msg.startswith("Unsupported color"):
            assert code == -5
        elif msg.startswith("Invalid pixel data length"):
            assert code == -6
        elif "exceeds maximum dimension" in msg:
            assert code == -4
        else:
            raise AssertionError(msg)
