"""Per-band entropy coding + splice on the CPU: the host twins of the band encoder
(`pixo_hip_jpeg_band_{count,bits,piece}_host`, product code in jpeg_host.cpp) and `pixo_hip_jpeg_splice`,
driven exactly like the multi-GPU path drives the device encoders — boundary DCs forward, bit totals
forward, pieces spliced — must reproduce the whole-image file of the oracle byte for byte, for every
number of bands (including more bands than MCU rows and bands of a few bits)."""
import numpy as np
import pytest

import oracle_lib as O
import synth
from pixo_amd import ColorType, jpeg


def banded_file(px, w, h, ct, ss, q, parts, optimize=False):
    o = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss)) \
        .optimize_huffman(optimize).build()
    bpp = 1 if ct == 0 else 3
    bands = [jpeg.band(w, h, ct, ss, parts, k) for k in range(parts)]
    tuples, lasts = [], []
    for b in bands:
        rows = b["row_end"] - b["row_begin"]
        if rows == 0:
            tuples.append(None); lasts.append(None)
            continue
        sub = px[b["row_begin"] * w * bpp: b["row_end"] * w * bpp]
        y, cb, cr = O.coeffs(sub, w, rows, ct, ss, q)
        tuples.append((y, cb, cr, rows))
        lasts.append([int(y[-1, 0]), int(cb[-1, 0]) if cb.shape[0] else 0, int(cr[-1, 0]) if cr.shape[0] else 0])
    prevs, prev = [], [0, 0, 0]
    for k in range(parts):
        prevs.append(list(prev))
        if lasts[k] is not None:
            prev = lasts[k]
    total = None
    if optimize:
        total = np.zeros(jpeg.COUNT_WORDS, np.uint64)
        for k, t in enumerate(tuples):
            if t is not None:
                total += jpeg.band_count_host(t[0], t[1], t[2], o, t[3], prevs[k])
    bits = [jpeg.band_bits_host(t[0], t[1], t[2], o, t[3], prevs[k], total) if t is not None else 0 for k, t in enumerate(tuples)]
    pieces, off = [], 0
    for k, t in enumerate(tuples):
        pieces.append(jpeg.band_piece_host(t[0], t[1], t[2], o, t[3], prevs[k], off, total) if t is not None else bytes(16))
        off += bits[k]
    return jpeg.splice(o, pieces, total), bits


@pytest.mark.parametrize("parts", [1, 2, 3, 5, 8, 13, 40])
@pytest.mark.parametrize("case", [(200, 203, 2, 1, 75), (97, 61, 2, 0, 90), (64, 100, 0, 0, 50)])
def test_spliced_bands_equal_the_whole_file(case, parts):
    w, h, ct, ss, q = case
    px = synth.noise_gray(w, h, 5) if ct == 0 else synth.noise(w, h, 5)
    got, bits = banded_file(px, w, h, ct, ss, q, parts)
    assert got == O.encode(px, O.make_options(w, h, ct, q, ss))
    assert sum(bits) > 0


@pytest.mark.parametrize("parts", [2, 3, 7])
def test_spliced_bands_with_optimised_tables(parts):
    w, h, ct, ss, q = 200, 203, 2, 1, 75
    px = synth.noise(w, h, 6)
    got, _ = banded_file(px, w, h, ct, ss, q, parts, optimize=True)
    assert got == O.encode(px, O.make_options(w, h, ct, q, ss, optimize_huffman=True))


def test_bands_of_a_few_bits_share_bytes():
    """A flat gray image: every block is six bits (DC category 0 + end of block), so with one block per
    band several bands end inside the same byte and most pieces consist of head or tail bits only."""
    w, h = 8, 8 * 24
    px = np.full(w * h, 128, np.uint8)
    for parts in (3, 8, 24):
        got, bits = banded_file(px, w, h, 0, 0, 80, parts)
        assert got == O.encode(px, O.make_options(w, h, 0, 80, 0))
    assert bits == [6] * 24


def test_0xff_bytes_across_band_boundaries_are_stuffed():
    """Saturated noise at q=100 produces many 0xFF bytes; with many bands some of them are shared bytes
    (tail of one band + head of the next), which only the splice can stuff."""
    w, h = 48, 16 * 37
    px = synth.extremes(w, h, 11)
    for parts in (2, 9, 37):
        got, _ = banded_file(px, w, h, 2, 1, 100, parts)
        assert got == O.encode(px, O.make_options(w, h, 2, 100, 1))


def test_splice_rejects_pieces_that_do_not_fit():
    w, h = 64, 64
    px = synth.noise(w, h, 1)
    o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420).build()
    y, cb, cr = O.coeffs(px, w, h, 2, 1, 80)
    good = jpeg.band_piece_host(y, cb, cr, o, h, [0, 0, 0], 0)
    assert jpeg.splice(o, [good]) == O.encode(px, O.make_options(w, h, 2, 80, 1))
    shifted = jpeg.band_piece_host(y, cb, cr, o, h, [0, 0, 0], 3)  # claims to start at bit 3 of the scan
    with pytest.raises(Exception, match="bit offset"):
        jpeg.splice(o, [good, shifted])
    with pytest.raises(Exception, match="malformed"):
        jpeg.splice(o, [good[:10]])


def test_layout_and_finish_place_every_byte_like_the_one_step_splice():
    """pixo_hip_jpeg_splice_layout / _finish: the bodies are copied to the offsets the layout names (what every
    GPU does with its own band), the rest is written by _finish — same file as pixo_hip_jpeg_splice."""
    w, h, parts = 120, 16 * 11, 5
    px = synth.extremes(w, h, 3)
    o = jpeg.JpegOptions.builder(w, h).quality(100).subsampling(jpeg.Subsampling.S420).build()
    bands = [jpeg.band(w, h, 2, 1, parts, k) for k in range(parts)]
    pieces, prev, off = [], [0, 0, 0], 0
    for b in bands:
        rows = b["row_end"] - b["row_begin"]
        y, cb, cr = O.coeffs(px[b["row_begin"] * w * 3: b["row_end"] * w * 3], w, rows, 2, 1, 100)
        pieces.append(jpeg.band_piece_host(y, cb, cr, o, rows, prev, off))
        off += jpeg.band_bits_host(y, cb, cr, o, rows, prev)
        prev = [int(y[-1, 0]), int(cb[-1, 0]), int(cr[-1, 0])]
    headers = [p[:16] for p in pieces]
    file_len, body_off = jpeg.splice_layout(o, headers)
    buf = np.full(file_len, 0xA5, np.uint8)
    for p, at in zip(pieces, body_off):
        buf[at: at + len(p) - 16] = np.frombuffer(p[16:], np.uint8)
    jpeg.splice_finish(o, headers, buf, file_len)
    want = O.encode(px, O.make_options(w, h, 2, 100, 1))
    assert buf.tobytes() == want == jpeg.splice(o, pieces)
    with pytest.raises(Exception, match="too small"):
        jpeg.splice_finish(o, headers, buf, file_len - 1)
