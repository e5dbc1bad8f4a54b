"""Checks that do not go through our own restatement of the reference: an independent decoder reads what we wrote.

* Restart intervals are unreachable through the reference's wasm entry (`encode_jpeg` has no such argument), so
  `src/jpeg/mod.rs:1423-1445` is pinned only by the oracle's restatement + marker structure.  Here Pillow (libjpeg)
  decodes the file WITH restart markers and the file WITHOUT them made from the same pixels: the coefficients are the
  same, so the decoded pixels must be identical — a wrong DC predictor reset, a misplaced marker, a wrong RSTn cycle
  or broken padding in front of a marker all change or break the decode.
* The reference's marker-walk tests (tests/jpeg_conformance.rs:505-654), transliterated, on the product's host coder.
* PNG strategies the wasm build cannot reach as standalone strategies (MinSum, the five fixed filters, stateless
  AdaptiveFast: vectors exist only for presets 0/1/2 = stateful AdaptiveFast, Adaptive, Bigrams): the filtered stream is
  wrapped into a real PNG (zlib + chunks) and Pillow must reconstruct the original pixels from it, whatever filter
  each row chose.
"""
import io
import struct
import zlib

import numpy as np
import pytest
from PIL import Image

import oracle_lib as O
import synth
from pixo_amd import ColorType, jpeg


def _decode(blob):
    im = Image.open(io.BytesIO(blob))
    im.load()
    return np.asarray(im)


@pytest.mark.parametrize("interval", [1, 2, 3, 7, 8, 9, 64])
@pytest.mark.parametrize("mode", [(2, 1), (2, 0), (0, 0)])
def test_restart_file_decodes_to_the_same_pixels_as_the_plain_file(interval, mode):
    ct, ss = mode
    w, h = 83, 61
    px = synth.noise_gray(w, h, 21) if ct == 0 else synth.gradient_rgb(w, h) ^ synth.noise(w, h, 21) >> 3
    y, cb, cr = O.coeffs(px, w, h, ct, ss, 85)
    b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(85).subsampling(jpeg.Subsampling(ss))
    plain = jpeg.entropy_encode(y, cb, cr, b.build())                       # the product's host coder
    with_rst = jpeg.entropy_encode(y, cb, cr, b.restart_interval(interval).build())
    assert with_rst == O.encode(px, O.make_options(w, h, ct, 85, ss, restart=interval))
    a, r = _decode(plain), _decode(with_rst)
    assert a.shape == r.shape == ((h, w) if ct == 0 else (h, w, 3))
    assert np.array_equal(a, r)
    # the markers really are there (otherwise the comparison proves nothing)
    unit = 16 if (ct == 2 and ss == 1) else 8
    units = ((w + unit - 1) // unit) * ((h + unit - 1) // unit)
    assert sum(1 for k in range(len(with_rst) - 1) if with_rst[k] == 0xFF and 0xD0 <= with_rst[k + 1] <= 0xD7) == (units - 1) // interval


def _walk_markers(blob):
    """tests/jpeg_conformance.rs:523-567: every segment up to SOS, lengths consistent."""
    assert blob[:2] == b"\xff\xd8" and blob[-2:] == b"\xff\xd9"
    off, seen = 2, []
    while off + 4 <= len(blob):
        assert blob[off] == 0xFF, "marker sync lost at %d" % off
        marker = blob[off + 1]
        off += 2
        if marker == 0xD9:
            break
        length = struct.unpack(">H", blob[off:off + 2])[0]
        assert length >= 2 and off + length <= len(blob)
        seen.append(marker)
        if marker == 0xDA:
            break
        off += length
    return seen


def test_marker_structure_with_and_without_restart_interval():
    # jpeg_marker_structure_with_restart_interval (:505-574): 16x12 4:2:0 q85 restart 4
    w, h = 16, 12
    px = synth.noise(w, h, 6262)
    y, cb, cr = O.coeffs(px, w, h, 2, 1, 85)
    o = jpeg.JpegOptions.builder(w, h).quality(85).subsampling(jpeg.Subsampling.S420).restart_interval(4).build()
    seen = _walk_markers(jpeg.entropy_encode(y, cb, cr, o))
    for m in (0xE0, 0xDB, 0xC0, 0xC4, 0xDD, 0xDA):
        assert m in seen
    # jpeg_no_restart_marker_without_interval (:576-592)
    w, h = 12, 9
    px = synth.noise(w, h, 7373)
    y, cb, cr = O.coeffs(px, w, h, 2, 0, 80)
    blob = jpeg.entropy_encode(y, cb, cr, jpeg.JpegOptions.builder(w, h).quality(80).build())
    assert b"\xff\xdd" not in blob[:blob.index(b"\xff\xda")] and 0xDD not in _walk_markers(blob)


@pytest.mark.parametrize("case", [(16, 16, 0, 4), (32, 32, 1, 2)])
def test_no_trailing_restart_marker_when_the_mcus_divide_evenly(case):
    # jpeg_no_trailing_restart_marker_when_divisible_444 / _420_exact_multiple (:594-654)
    w, h, ss, interval = case
    px = synth.noise(w, h, 9999)
    y, cb, cr = O.coeffs(px, w, h, 2, ss, 85)
    o = jpeg.JpegOptions.builder(w, h).quality(85).subsampling(jpeg.Subsampling(ss)).restart_interval(interval).build()
    blob = jpeg.entropy_encode(y, cb, cr, o)
    assert blob.endswith(b"\xff\xd9")
    assert not (blob[-4] == 0xFF and 0xD0 <= blob[-3] <= 0xD7)
    assert _decode(blob).shape == (h, w, 3)


# ---- PNG: an independent decoder undoes the row filters ------------------------------------------------------------
def _png_from_filtered(flt, w, h, color_type, adler):
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    z = zlib.compress(flt.tobytes(), 1)
    assert struct.unpack(">I", z[-4:])[0] == adler  # the zlib trailer IS the checksum our stage delivers
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0)) + chunk(b"IDAT", z) + chunk(b"IEND", b"")


@pytest.mark.parametrize("strategy", [O.S_NONE, O.S_SUB, O.S_UP, O.S_AVERAGE, O.S_PAETH, O.S_MINSUM, O.S_ADAPTIVE, O.S_ADAPTIVE_FAST,
                                      O.S_BIGRAMS])
@pytest.mark.parametrize("bpp", [1, 2, 3, 4])
def test_filtered_stream_is_a_png_that_pillow_reconstructs(strategy, bpp):
    """MinSum, stateless AdaptiveFast and the fixed filters cannot be produced by the reference's wasm build on their own:
    here the proof is the PNG definition itself — any conforming decoder must get the pixels back."""
    w, h = 97, 45
    px = (synth.lcg_bytes(w * h * bpp, 5 + strategy) & 0xF8) | (synth.gradient_rgb(w * bpp, h)[: w * h * bpp] >> 5)
    flt, adler = O.png_filter(px, w, h, bpp, strategy)
    png = _png_from_filtered(flt, w, h, {1: 0, 2: 4, 3: 2, 4: 6}[bpp], adler)
    got = _decode(png)
    assert np.array_equal(got.reshape(-1), px)
    assert set(flt.reshape(h, w * bpp + 1)[:, 0]) <= {0, 1, 2, 3, 4}
