"""Reference-made vectors on PHOTOGRAPH-LIKE content (tests/golden/jpeg_scene_cases.json, made by tests/golden/make_golden_scene.py
with the reference's own wasm build): `synth.photo` and `synth.scene` — the content class the reference benchmarks
(benches/BENCHMARKS.md:92-93: real photographs, which cannot travel) — presets 0 / 1 / 2, both subsamplings, gray.  The CPU oracle
must reproduce every one byte for byte (CPU test); the HIP library as well (GPU test, through the wasm-shaped flat entry)."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

SMALL = G.scene_cases(max_pixels=1100 * 1100)
LARGE = G.scene_cases(min_pixels=1100 * 1100 + 1)


@pytest.mark.parametrize("c", SMALL, ids=[c["name"] for c in SMALL])
def test_oracle_matches_reference_on_photo_like_content(c):
    G.check(c, O.encode_flat(G.make_input(c), c["w"], c["h"], c["color_type"], c["quality"], c["preset"], c["s420"]))


@pytest.mark.slow
@pytest.mark.parametrize("c", LARGE, ids=[c["name"] for c in LARGE])
def test_oracle_matches_reference_on_large_photo_like_content(c):
    G.check(c, O.encode_flat(G.make_input(c), c["w"], c["h"], c["color_type"], c["quality"], c["preset"], c["s420"]))


@pytest.mark.gpu
@pytest.mark.parametrize("c", SMALL + LARGE, ids=[c["name"] for c in SMALL + LARGE])
def test_hip_library_matches_reference_on_photo_like_content(c):
    from pixo_amd import jpeg
    px = np.ascontiguousarray(G.make_input(c))
    G.check(c, bytes(jpeg.encode_jpeg(px, c["w"], c["h"], c["color_type"], c["quality"], c["preset"], c["s420"])))
    if c["preset"] == 0 and c["color_type"] == 2:  # device pixels: the fused pixel -> scan kernel
        import torch
        o = jpeg.JpegOptions.builder(c["w"], c["h"]).quality(c["quality"]).subsampling(jpeg.Subsampling(1 if c["s420"] else 0)).build()
        G.check(c, jpeg.encode_device(torch.from_numpy(px).cuda(), o))
    assert jpeg.lookback_fallbacks() == 0
