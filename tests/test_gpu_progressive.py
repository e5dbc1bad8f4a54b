"""Preset 2 ("max": optimised tables + progressive scans + trellis quantisation) through the HIP
backend: raw-mode coefficient kernel -> trellis kernel -> host progressive coder.  Byte-identical to
the files the reference's wasm build made (tests/golden/make_golden_p2.py) and to the oracle.  -m gpu."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
import synth
from pixo_amd import ColorType, jpeg

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG  # noqa: E402

CASES = json.load(open(os.path.join(HERE, "golden", "jpeg_p2_cases.json")))["cases"]


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_reference_made_preset2_files(c):
    px = MG.GEN[c["gen"]](c["w"], c["h"], c["seed"])
    got = jpeg.encode_jpeg(px, c["w"], c["h"], c["color_type"], c["quality"], 2, c["s420"])
    assert len(got) == c["len"] and hashlib.sha256(got).hexdigest() == c["sha256"]


def _opts(w, h, ct, ss, q, **kw):
    b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss))
    for k, v in kw.items():
        b = getattr(b, k)(v)
    return b.build()


@pytest.mark.parametrize("mode", [(2, 1), (2, 0), (0, 0)])
@pytest.mark.parametrize("flags", [(False, False), (True, False), (False, True), (True, True)])
def test_every_progressive_combination_against_the_oracle(mode, flags):
    ct, ss = mode
    trellis, optimize = flags
    for (w, h, q, kind) in [(200, 120, 80, 0), (333, 77, 35, 0), (520, 40, 95, 1), (64, 64, 100, 0), (9, 300, 60, 0)]:
        if ct == 2:
            px = synth.noise(w, h, q) if kind == 0 else synth.gradient_rgb(w, h)
        else:
            px = synth.noise_gray(w, h, q) if kind == 0 else synth.gradient_rgb(w, h).reshape(-1, 3)[:, 1].copy()
        got = jpeg.encode(px, _opts(w, h, ct, ss, q, progressive=True, trellis_quant=trellis, optimize_huffman=optimize))
        want = O.encode(px, O.make_options(w, h, ct, q, ss, progressive=True, trellis=trellis, optimize_huffman=optimize))
        assert got == want, (w, h, q, kind)


def test_progressive_device_entry_points_and_restart_interval_header():
    import torch
    w, h = 300, 200
    px = synth.noise(w, h, 12)
    d_px = torch.from_numpy(px).to("cuda:0")
    torch.cuda.synchronize()
    o = _opts(w, h, 2, 1, 75, progressive=True, trellis_quant=True, optimize_huffman=True, restart_interval=7)
    want = O.encode(px, O.make_options(w, h, 2, 75, 1, progressive=True, trellis=True, optimize_huffman=True, restart=7))
    assert jpeg.encode_device(d_px, o) == want
    assert jpeg.encode_batch_device(torch.cat([d_px, d_px]), o, 2) == [want, want]
