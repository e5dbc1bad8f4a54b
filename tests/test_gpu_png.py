"""GPU parity of the PNG row-filter stage (pixo_hip_png_filter*, png_filter.hip) against the
oracle and against the vectors made by the reference's own wasm build: filtered stream bytes and
Adler-32, bit-exact.  -m gpu."""
import hashlib
import json
import os
import sys
import zlib

import numpy as np
import pytest

import oracle_lib as O
from pixo_amd import png

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_png as MG  # noqa: E402
import synth  # noqa: E402

CASES = json.load(open(os.path.join(HERE, "golden", "png_cases.json")))["cases"]
BPP = {0: 1, 1: 2, 2: 3, 3: 4}


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_reference_made_vectors(c):
    px = MG.make_input(c)
    bpp = BPP[c["color_type"]]
    # preset 0: AdaptiveFast of the wasm build (no rayon -> sequential, stateful); 1: Adaptive; 2: Bigrams
    strategy, flags = {0: (png.FilterStrategy.ADAPTIVE_FAST, png.NO_RAYON), 1: (png.FilterStrategy.ADAPTIVE, 0),
                       2: (png.FilterStrategy.BIGRAMS, 0)}[c["preset"]]
    flt, adler = png.apply_filters(px, c["w"], c["h"], bpp, strategy, flags)
    row = c["w"] * bpp + 1
    assert "".join(str(int(f)) for f in flt[::row]) == c["filters"]
    assert hashlib.sha256(flt.tobytes()).hexdigest() == c["filtered_sha256"]
    assert adler == c["adler32"]


@pytest.mark.parametrize("bpp", [1, 2, 3, 4, 6, 8])
@pytest.mark.parametrize("strategy", [0, 1, 2, 3, 4, 5, 6, 7, 8])
def test_every_strategy_and_pixel_size_against_the_oracle(bpp, strategy):
    for (w, h, seed) in [(67, 41, 1), (256, 40, 2), (1000, 35, 3), (5, 900, 4), (1, 70, 5), (4100, 33, 6)]:
        px = synth.lcg_bytes(w * h * bpp, seed + bpp)
        if seed % 2 == 0:  # smoother content: other filters win
            px = (np.cumsum(px.astype(np.int64) % 5) % 256).astype(np.uint8)
        want, wad = O.png_filter(px, w, h, bpp, strategy, stateful_fast=False)
        got, gad = png.apply_filters(px, w, h, bpp, strategy)
        assert np.array_equal(got, want), (w, h, seed)
        assert gad == wad == zlib.adler32(want.tobytes())


def test_small_images_and_sequential_adaptive_fast():
    # <= 4096 pixels: adaptive strategies become Sub; height <= 32: AdaptiveFast is the stateful variant
    for (w, h) in [(64, 64), (10, 3), (300, 32), (300, 33), (2000, 2)]:
        px = synth.lcg_bytes(w * h * 4, w)
        for strategy in (6, 7, 8):
            want, wad = O.png_filter(px, w, h, 4, strategy, stateful_fast=(h <= 32))
            got, gad = png.apply_filters(px, w, h, 4, strategy)
            assert np.array_equal(got, want) and gad == wad, (w, h, strategy)


def test_device_pointers_unaligned_rows_and_errors():
    import torch
    w, h, bpp = 333, 50, 3  # 999-byte rows: byte-assembled loads
    px = synth.lcg_bytes(w * h * bpp, 8)
    d_in = torch.from_numpy(px).to("cuda:0")
    d_out = torch.empty(png.filtered_size(w, h, bpp), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    adler = png.apply_filters_device(d_in, w, h, bpp, d_out, png.FilterStrategy.ADAPTIVE)
    want, wad = O.png_filter(px, w, h, bpp, O.S_ADAPTIVE)
    assert np.array_equal(d_out.cpu().numpy(), want) and adler == wad
    from pixo_amd import Error
    with pytest.raises(Error):
        png.apply_filters(px, w, h, bpp, 9)  # no such strategy
    with pytest.raises(Error):
        png.apply_filters(px[:-1], w, h, bpp)
    with pytest.raises(Error):
        png.apply_filters(px, w, h, 5)


def test_bigrams_on_long_rows_structured_content_and_the_c5_shape():
    """Bigrams (preset "max"): rows longer than the LDS stage (direct stores), rows whose byte pairs
    repeat heavily (few distinct pairs: ties between filters decide), a 2-byte row (one pair), a
    1-byte row (no pair at all: every score 0, None wins) and the 4096x4096 RGBA shape of config 5."""
    rng = np.random.RandomState(3)
    shapes = [(20000, 34, 4), (2, 2100, 1), (1, 5000, 1), (700, 40, 3), (4096, 64, 4)]
    for (w, h, bpp) in shapes:
        for kind in range(3):
            if kind == 0:
                px = synth.lcg_bytes(w * h * bpp, w + kind)
            elif kind == 1:
                px = np.tile(rng.randint(0, 256, 7).astype(np.uint8), w * h * bpp // 7 + 1)[: w * h * bpp].copy()
            else:
                px = (np.cumsum(synth.lcg_bytes(w * h * bpp, 9).astype(np.int64) % 3) % 256).astype(np.uint8)
            want, wad = O.png_filter(px, w, h, bpp, O.S_BIGRAMS)
            got, gad = png.apply_filters(px, w, h, bpp, png.FilterStrategy.BIGRAMS)
            assert np.array_equal(got, want) and gad == wad, (w, h, bpp, kind)
    w = h = 4096
    px = synth.lcg_bytes(w * h * 4, 42)
    want, wad = O.png_filter(px, w, h, 4, O.S_BIGRAMS)
    got, gad = png.apply_filters(px, w, h, 4, png.FilterStrategy.BIGRAMS)
    assert hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(want.tobytes()).hexdigest() and gad == wad


@pytest.mark.parametrize("strategy", [5, 6, 7])
def test_rows_around_the_register_path_limit(strategy):
    """The adaptive strategies hold rows of at most 16 KiB (256 threads) or 32 KiB (512 threads) in registers
    (K_REGS) and take the two-pass form beyond: rows around both limits, plus short rows whose last group is
    partial, all against the oracle."""
    for (w, bpp) in [(4095, 4), (4096, 4), (4097, 4), (4100, 4), (5461, 3), (5462, 3), (16384, 1), (16385, 1), (2731, 6), (37, 1), (3, 2),
                     (8191, 4), (8192, 4), (8193, 4), (5000, 6), (4096, 8), (4097, 8), (32768, 1), (32769, 1)]:  # 512-thread form and beyond
        h = 34
        px = synth.lcg_bytes(w * h * bpp, w)
        px[w * bpp * 7: w * bpp * 9] = (np.cumsum(px[w * bpp * 7: w * bpp * 9].astype(np.int64) % 3) % 256).astype(np.uint8)
        want, wad = O.png_filter(px, w, h, bpp, strategy, stateful_fast=False)
        got, gad = png.apply_filters(px, w, h, bpp, strategy)
        assert np.array_equal(got, want) and gad == wad, (w, bpp)


def test_paeth_predictor_on_every_left_up_upleft_combination():
    """All 2^24 (left, up, up-left) byte triples: 8192 rows of 12288 one-byte pixels, row pair p holding 4096 of
    them at positions 3k + 1 (even row: up-left, up; odd row: left, sample).  Fixed Paeth through the device
    predictor (the same packed 16-bit function the adaptive strategies score with) against the oracle."""
    pairs, per = 4096, 4096
    w, h = 3 * per, 2 * pairs
    rng = np.random.RandomState(12)
    img = rng.randint(0, 256, (h, w)).astype(np.uint8)
    t = (np.arange(pairs, dtype=np.uint32)[:, None] * per + np.arange(per, dtype=np.uint32)[None, :])
    img[0::2, 0::3] = (t >> 16).astype(np.uint8)          # up-left
    img[0::2, 1::3] = ((t >> 8) & 255).astype(np.uint8)   # up
    img[1::2, 0::3] = (t & 255).astype(np.uint8)          # left
    px = img.reshape(-1)
    want, wad = O.png_filter(px, w, h, 1, O.S_PAETH)
    got, gad = png.apply_filters(px, w, h, 1, png.FilterStrategy.PAETH)
    assert np.array_equal(got, want) and gad == wad
    got, gad = png.apply_filters(px, w, h, 1, png.FilterStrategy.ADAPTIVE)
    want, wad = O.png_filter(px, w, h, 1, O.S_ADAPTIVE)
    assert np.array_equal(got, want) and gad == wad


@pytest.mark.parametrize("strategy", list(range(9)))
def test_gpu_filtered_stream_is_a_png_that_pillow_reconstructs(strategy):
    """MinSum, the five fixed filters and the stateless AdaptiveFast cannot be produced by the reference's wasm build as
    standalone strategies (vectors exist for presets 0/1/2 only): for those the independent check is the PNG format
    itself — the device's filtered stream, wrapped into a PNG, must give Pillow the original pixels back."""
    import io
    import struct
    import zlib
    from PIL import Image
    for bpp, color_type in ((3, 2), (4, 6), (1, 0)):
        w, h = 301, 77
        px = (synth.lcg_bytes(w * h * bpp, 9 + strategy) & 0xF8) | (synth.gradient_rgb(w * bpp, h)[: w * h * bpp] >> 5)
        flt, adler = png.apply_filters(px, w, h, bpp, png.FilterStrategy(strategy))
        z = zlib.compress(flt.tobytes(), 1)
        assert struct.unpack(">I", z[-4:])[0] == adler

        def chunk(tag, data):
            return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
        blob = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0)) + chunk(b"IDAT", z) + chunk(b"IEND", b"")
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(blob))).reshape(-1), px)


def test_config_5_full_size_known_answer_of_the_reference():
    """SURVEY §8c's C5 known answer, made by the reference's own wasm build: 4096x4096 RGBA LCG bytes (seed 42, every 4th byte | 1),
    FilterStrategy::Adaptive -> filtered stream 67,112,960 B sha256 240e005d..., Adler-32 0x90cc12e3, per-row filter histogram
    {None 1316, Sub 800, Up 436, Avg 1048, Paeth 496}.  (bench.py refuses to report C5 without it; VERDICT r5 asked for it as a
    test as well.)  Host pixels and device pixels."""
    import torch
    w = h = 4096
    px = synth.rgba_noise_alpha1(w, h, 42)
    got, adler = png.apply_filters(px, w, h, 4, png.FilterStrategy.ADAPTIVE)
    assert got.size == 67112960 and adler == 0x90CC12E3
    assert hashlib.sha256(got.tobytes()).hexdigest() == "240e005d4da54561ff45b86d92d482cf81b44c78add39fcb6e3f5a600edbd2b0"
    hist = np.bincount(got.reshape(h, w * 4 + 1)[:, 0], minlength=5)
    assert hist.tolist() == [1316, 800, 436, 1048, 496]
    d = torch.from_numpy(px).cuda()
    out = torch.empty(67112960, dtype=torch.uint8, device="cuda")
    assert png.apply_filters_device(d, w, h, 4, out, png.FilterStrategy.ADAPTIVE) == 0x90CC12E3
    assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == "240e005d4da54561ff45b86d92d482cf81b44c78add39fcb6e3f5a600edbd2b0"
