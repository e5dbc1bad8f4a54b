"""The DEVICE tile code (pixo_amd/csrc/jpeg_tile.h), compiled for the host and driven lane by
lane (tests/emu: producer wavefront, then the three consumer wavefronts), against the oracle — bit-exact.  This exercises on CPU everything about the
kernel except the hardware itself: lane->pixel/block mapping, LDS layout and swizzles, packed
u16 colour math, DCT operation order, the quantiser fast path and its exact fallback, edge
replication, the interior/edge tile split, the dword-aligned loads and the funnel loads of unaligned rows."""
import numpy as np
import pytest

import emu_lib as E
import golden_util as G
import oracle_lib as O
import synth


def _same(px, w, h, ct, ss, q, **kw):
    oy, ocb, ocr = O.coeffs(px, w, h, ct, ss, q)
    ey, ecb, ecr, stats = E.coeffs(px, w, h, ct, ss, q, **kw)
    assert np.array_equal(oy, ey)
    assert np.array_equal(ocb, ecb) and np.array_equal(ocr, ecr)
    return stats


SMALL = [c for c in G.cases(max_pixels=520 * 520) if c["preset"] == 0]


@pytest.mark.parametrize("c", SMALL, ids=[c["name"] for c in SMALL])
def test_emulated_kernel_on_golden_inputs(c):
    _same(G.make_input(c), c["w"], c["h"], c["color_type"], 1 if c["s420"] else 0, c["quality"])


@pytest.mark.parametrize("w,h", [(512, 16), (1024, 32), (1536, 48), (2048, 16), (516, 20), (1028, 33),
                                 (511, 16), (513, 17), (1030, 40), (4, 4), (12, 300)])
@pytest.mark.parametrize("mode", [(2, 1), (2, 0), (0, 0)])
def test_interior_and_edge_tiles(w, h, mode):
    ct, ss = mode
    px = synth.noise_gray(w, h, w + h) if ct == 0 else synth.noise(w, h, w + h)
    aligned, gather, funnel = _same(px, w, h, ct, ss, 80)
    row_bytes = w * (3 if ct == 2 else 1)
    # dword-aligned rows -> every tile via the 12-byte loads, including the right/bottom edge tiles;
    # other rows -> aligned dwords + alignbyte (funnel); the byte gather only below one 4-pixel group
    if row_bytes % 4 == 0 and w >= 4:
        assert aligned > 0 and gather == 0 and funnel == 0
    else:
        assert aligned == 0 and gather == 0 and funnel > 0


@pytest.mark.parametrize("w,h", [(1, 1), (2, 9), (3, 17), (3, 3)])
@pytest.mark.parametrize("mode", [(2, 1), (2, 0), (0, 0)])
def test_images_narrower_than_a_group_take_the_byte_gather(w, h, mode):
    ct, ss = mode
    px = synth.noise_gray(w, h, 5) if ct == 0 else synth.noise(w, h, 5)
    aligned, gather, funnel = _same(px, w, h, ct, ss, 80)
    assert gather > 0 and aligned == 0 and funnel == 0


@pytest.mark.parametrize("w", [4, 5, 6, 7, 9, 510, 511, 513, 514, 515, 1021, 1022, 1023, 1025, 1366, 1918])
@pytest.mark.parametrize("mode", [(2, 1), (2, 0), (0, 0)])
def test_funnel_loads_every_width_residue_and_base_alignment(w, mode):
    """Rows that are not dword aligned: aligned dwords + v_alignbyte, the partial last group (1-3 valid
    pixels) rebuilt in registers, every base alignment; also forced on aligned images."""
    ct, ss = mode
    h = 19
    px = synth.noise_gray(w, h, w) if ct == 0 else synth.noise(w, h, w)
    for mis in (0, 1, 2, 3):
        st = _same(px, w, h, ct, ss, 80, misalign=mis)
        assert st[1] == 0
    _same(px, w, h, ct, ss, 80, allow_fast=2)


@pytest.mark.parametrize("mode", [(2, 1), (2, 0), (0, 0)])
def test_unaligned_base_and_forced_gather_path_agree(mode):
    ct, ss = mode
    w, h = 1024, 48
    px = synth.noise_gray(w, h, 1) if ct == 0 else synth.noise(w, h, 1)
    _same(px, w, h, ct, ss, 75, allow_fast=False)
    for mis in (1, 2, 3):
        st = _same(px, w, h, ct, ss, 75, misalign=mis)
        assert st[0] == 0 and st[2] > 0  # misaligned base: never the 12-byte loads, always the funnel


@pytest.mark.parametrize("q", [1, 5, 20, 49, 50, 77, 90, 100])
def test_all_quality_branches(q):
    _same(synth.noise(96, 48, q), 96, 48, 2, q % 2, q)
    _same(synth.gradient_rgb(96, 48), 96, 48, 2, 1, q)


def test_saturated_colours_hit_the_chroma_clamp():
    # pure blue -> Cb 256 before clamp; pure red -> Cr 256 (color.rs:73-76)
    for rgb in ([0, 0, 255], [255, 0, 0], [255, 255, 255], [0, 0, 0], [0, 255, 0], [1, 0, 255]):
        px = np.tile(np.array(rgb, np.uint8), 64 * 32)
        _same(px, 64, 32, 2, 1, 100)
        _same(px, 64, 32, 2, 0, 100)
    for seed in (1, 2, 3):  # clamped and unclamped pixels inside the same 2x2 box / block
        _same(synth.extremes(80, 48, seed), 80, 48, 2, 1, 100)
        _same(synth.extremes(80, 48, seed), 80, 48, 2, 0, 90)


@pytest.mark.parametrize("mode", [(2, 1), (2, 0), (0, 0)])
@pytest.mark.parametrize("order", [0, 1, 2])
def test_consumer_wavefronts_are_independent(mode, order):
    """The three consumer wavefronts share no LDS they write (own stage region, read-only planar
    buffer): running them in any order, each one completely (rows, columns+quantise, store),
    must not change a single coefficient."""
    ct, ss = mode
    w, h = 1100, 70
    px = synth.noise_gray(w, h, 3) if ct == 0 else synth.noise(w, h, 3)
    _same(px, w, h, ct, ss, 80, wave_order=order)
