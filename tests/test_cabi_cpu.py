"""CPU-only checks of the C-ABI library and the host side of the path: symbols, option
semantics (presets/builder), validation order and messages (recorded from the reference),
tuple geometry, band sharding and the product's host entropy coder against the goldens.
No GPU compute is called here."""
import ctypes as C

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
import synth
from pixo_amd import ColorType, _lib, error, jpeg


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    for name in _lib.SYMBOLS:
        assert hasattr(lib, name), name
    # header and binding agree on the symbol list
    import os, re
    hdr = open(os.path.join(os.path.dirname(_lib.__file__), "..", "include", "pixo_hip.h")).read()
    declared = set(re.findall(r"\b(pixo_(?:hip|jpeg)_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS)
    assert b"gfx950" in lib.pixo_hip_version()


def test_options_struct_layout_matches_header():
    assert C.sizeof(_lib.JpegOptionsC) == 20  # 2*u32 + 4*u8 + u16 + 3*u8 + pad
    o = _lib.JpegOptionsC()
    _lib.load().pixo_jpeg_options_from_preset(C.byref(o), 7, 9, 66, 2)
    assert (o.width, o.height, o.quality, o.color_type) == (7, 9, 66, 2)
    assert (o.subsampling, o.optimize_huffman, o.progressive, o.trellis_quant) == (1, 1, 1, 1)


def test_presets_and_builder_follow_reference():
    # jpeg/mod.rs:162-216
    f = jpeg.JpegOptions.fast(3, 4, 60)
    assert (f.subsampling, f.optimize_huffman, f.progressive, f.trellis_quant) == (jpeg.Subsampling.S444, False, False, False)
    b = jpeg.JpegOptions.balanced(3, 4, 60)
    assert (b.subsampling, b.optimize_huffman, b.progressive) == (jpeg.Subsampling.S444, True, False)
    m = jpeg.JpegOptions.max(3, 4, 60)
    assert (m.subsampling, m.optimize_huffman, m.progressive, m.trellis_quant) == (jpeg.Subsampling.S420, True, True, True)
    assert jpeg.JpegOptions.from_preset(1, 1, 50, 7).optimize_huffman  # other -> balanced
    d = jpeg.JpegOptions()
    assert (d.quality, d.subsampling, d.color_type, d.restart_interval) == (75, jpeg.Subsampling.S444, ColorType.Rgb, None)
    # .preset() keeps w/h/colour/quality (jpeg/mod.rs:285-293); later setters override
    o = jpeg.JpegOptions.builder(10, 20).color_type(ColorType.Gray).quality(33).preset(2) \
        .subsampling(jpeg.Subsampling.S444).build()
    assert (o.width, o.height, o.color_type, o.quality) == (10, 20, ColorType.Gray, 33)
    assert o.subsampling == jpeg.Subsampling.S444 and o.progressive and o.trellis_quant
    assert ColorType.Rgba.bytes_per_pixel() == 4 and ColorType.try_from(2) is ColorType.Rgb
    with pytest.raises(ValueError):
        ColorType.try_from(9)


ERRS = G.load()["errors"]


@pytest.mark.parametrize("e", ERRS, ids=[str(i) for i in range(len(ERRS))])
def test_flat_entry_errors_match_reference_strings(e):
    """Same 7-argument call the reference's wasm export received; identical Display text,
    reported before any device work (so this runs without a GPU)."""
    data = synth.lcg_bytes(e["nbytes"], 3)
    with pytest.raises(error.Error) as ei:
        jpeg.encode_jpeg(data, e["w"], e["h"], e["color_type"], e["quality"], e["preset"], e["s420"])
    assert str(ei.value) == e["error"]


def test_validation_order_and_variants():
    px = synth.noise(4, 4)
    B = jpeg.JpegOptions.builder
    with pytest.raises(error.InvalidQuality, match="Invalid quality 0: must be 1-100"):
        jpeg.encode(px, B(0, 0).quality(0).restart_interval(0).build())  # quality first
    with pytest.raises(error.InvalidRestartInterval, match=r"Invalid restart interval 0: must be 1-65535 \(or None to disable\)"):
        jpeg.encode(px, B(0, 0).quality(50).restart_interval(0).build())  # then restart
    with pytest.raises(error.InvalidDimensions, match="Invalid image dimensions: 0x4"):
        jpeg.encode(px, B(0, 4).build())
    with pytest.raises(error.ImageTooLarge, match="Image 65536x1 exceeds maximum dimension 65535"):
        jpeg.encode(px, B(65536, 1).color_type(ColorType.Rgba).build())  # size before colour
    with pytest.raises(error.UnsupportedColorType, match="Unsupported color type for this format"):
        jpeg.encode(px, B(4, 4).color_type(ColorType.Rgba).build())
    with pytest.raises(error.UnsupportedColorType):
        jpeg.encode(px, B(4, 4).color_type(ColorType.GrayAlpha).build())
    with pytest.raises(error.InvalidDataLength, match="Invalid pixel data length: expected 48 bytes, got 47"):
        jpeg.encode(px[:47], B(4, 4).build())
    # encode_into leaves the caller's buffer untouched on error
    out = bytearray(b"keep")
    with pytest.raises(error.InvalidQuality):
        jpeg.encode_into(out, px, B(4, 4).quality(101).build())
    assert out == b"keep"


def test_trellis_alone_is_the_baseline_encode():
    # the reference only reads trellis_quant inside its progressive path (jpeg/mod.rs:872-976):
    # with progressive off the flag changes nothing
    y = np.zeros((1, 64), np.int16)
    t = jpeg.JpegOptions.builder(8, 8).color_type(ColorType.Gray).trellis_quant(True).build()
    plain = jpeg.JpegOptions.builder(8, 8).color_type(ColorType.Gray).build()
    assert jpeg.entropy_encode(y, y, y, t) == jpeg.entropy_encode(y, y, y, plain)


@pytest.mark.parametrize("shape", [(72, 40, 2, 1), (50, 33, 2, 0), (31, 17, 0, 0), (300, 200, 2, 1), (640, 480, 2, 1)])
def test_host_progressive_coder_matches_oracle(shape):
    """The product's progressive scan coder (jpeg_host.cpp: SOF2, seven scans, end-of-band runs, the
    (0, 4) fallback for symbols the table lacks) against the oracle's restatement on the same tuple."""
    w, h, ct, ss = shape
    for gen, q in ((synth.noise, 80), (synth.gradient_rgb, 90), (synth.flat_blocks, 50)):
        px = gen(w, h) if gen is not synth.noise else synth.noise(w, h, 3)
        if ct == 0:
            px = px.reshape(-1, 3)[:, 0].copy()
        y, cb, cr = O.coeffs(px, w, h, ct, ss, q)
        o = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss)).progressive(True).build()
        got = jpeg.entropy_encode(y, cb, cr, o)
        want = O.encode_from_coeffs(y, cb, cr, O.make_options(w, h, ct, q, ss, progressive=True))
        assert got == want
        assert got[:2] == b"\xff\xd8" and b"\xff\xc2" in got[:700] and got.count(b"\xff\xda") >= 7


def test_no_gpu_means_loud_failure_not_a_cpu_fallback():
    if jpeg.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(error.CompressionError, match="no CPU fallback"):
        jpeg.encode(synth.noise(16, 16), jpeg.JpegOptions.builder(16, 16).build())


@pytest.mark.parametrize("w,h", [(1, 1), (8, 8), (9, 17), (16, 16), (17, 33), (1920, 1080), (4096, 4096), (65535, 3)])
def test_geometry_matches_oracle(w, h):
    for ct, ss in [(0, 0), (0, 1), (2, 0), (2, 1)]:
        assert jpeg.coefficient_geometry(w, h, ct, ss) == O.geometry(w, h, ct, ss)


def test_bands_partition_the_tuple():
    for (w, h, ct, ss) in [(4096, 4096, 2, 1), (1000, 999, 2, 1), (333, 77, 2, 0), (50, 9, 0, 0), (16384, 16384, 2, 1)]:
        yb, cb = jpeg.coefficient_geometry(w, h, ct, ss)
        for parts in (1, 2, 3, 8):
            yo = co = row = 0
            for i in range(parts):
                b = jpeg.band(w, h, ct, ss, parts, i)
                assert b["y_offset"] == yo and b["c_offset"] == co and b["row_begin"] == row
                # a band is an independent sub-image: its own geometry equals its share
                if b["row_end"] > b["row_begin"]:
                    g = jpeg.coefficient_geometry(w, b["row_end"] - b["row_begin"], ct, ss)
                    assert g == (b["y_blocks"], b["c_blocks"])
                yo += b["y_blocks"]; co += b["c_blocks"]; row = b["row_end"]
            assert (yo, co, row) == (yb, cb, h)


SMALL = G.cases(max_pixels=1100 * 1100)


@pytest.mark.parametrize("c", SMALL, ids=[c["name"] for c in SMALL])
def test_host_entropy_coder_reproduces_reference_files(c):
    """Product host code (headers, Huffman incl. optimised tables, bit writer, stuffing)
    fed with oracle coefficients must give the reference's exact bytes."""
    ss = 1 if c["s420"] else 0
    y, cb, cr = O.coeffs(G.make_input(c), c["w"], c["h"], c["color_type"], ss, c["quality"])
    o = jpeg.JpegOptions.builder(c["w"], c["h"]).color_type(ColorType(c["color_type"])) \
        .quality(c["quality"]).preset(c["preset"]).subsampling(jpeg.Subsampling(ss)).build()
    G.check(c, jpeg.entropy_encode(y, cb, cr, o))


@pytest.mark.parametrize("interval", [1, 2, 3, 7, 8, 9, 64, 65535])
@pytest.mark.parametrize("mode", [(2, 1), (2, 0), (0, 0)])
def test_restart_intervals_match_oracle_and_marker_structure(interval, mode):
    """restart_interval is unreachable through the reference's wasm entry, so restart parity
    rests on the oracle's restatement of jpeg/mod.rs:1423-1445 plus the structural rules the
    reference tests (tests/jpeg_conformance.rs:505-654): DRI present, RSTn cycle 0..7, none
    after the last MCU."""
    ct, ss = mode
    w, h = 50, 37
    px = synth.noise_gray(w, h, 11) if ct == 0 else synth.noise(w, h, 11)
    y, cb, cr = O.coeffs(px, w, h, ct, ss, 70)
    for opt_huff in (False, True):
        o = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(70) \
            .subsampling(jpeg.Subsampling(ss)).restart_interval(interval).optimize_huffman(opt_huff).build()
        got = jpeg.entropy_encode(y, cb, cr, o)
        want = O.encode(px, O.make_options(w, h, ct, 70, ss, restart=interval, optimize_huffman=opt_huff))
        assert got == want
        i = got.index(b"\xff\xdd")
        assert got[i:i + 6] == b"\xff\xdd\x00\x04" + interval.to_bytes(2, "big")
        sos = got.index(b"\xff\xda")
        scan = got[sos:]
        rst = [scan[k + 1] for k in range(len(scan) - 1) if scan[k] == 0xFF and 0xD0 <= scan[k + 1] <= 0xD7]
        units = (O.geometry(w, h, ct, ss)[1] if (ct == 2 and ss == 1) else O.geometry(w, h, ct, ss)[0])
        assert len(rst) == (units - 1) // interval
        assert rst == [0xD0 + (k % 8) for k in range(len(rst))]
        assert got.endswith(b"\xff\xd9") and not (scan[-4] == 0xFF and 0xD0 <= scan[-3] <= 0xD7)


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/pixo_hip.h is the FFI contract: it must compile as strict C99 (what cgo / bindgen / ctypesgen
    read) and a C program must link against the library and get the documented answers without a GPU."""
    import os, subprocess
    root = os.path.join(os.path.dirname(_lib.__file__), "..")
    src = tmp_path / "t.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "pixo_hip.h"
int main(void) {
    pixo_jpeg_options o;
    size_t yb = 0, cb = 0;
    pixo_jpeg_options_from_preset(&o, 4096, 4096, 80, 2);
    if (!(o.progressive && o.trellis_quant && o.optimize_huffman && o.subsampling == PIXO_S420)) return 1;
    if (pixo_hip_coeff_geometry(4096, 4096, PIXO_RGB, PIXO_S420, &yb, &cb) != PIXO_OK) return 2;
    if (yb != 262144 || cb != 65536) return 3;
    if (pixo_hip_coeff_geometry(0, 7, PIXO_RGB, PIXO_S420, &yb, &cb) != PIXO_ERR_INVALID_DIMENSIONS) return 4;
    if (strcmp(pixo_hip_last_error(), "Invalid image dimensions: 0x7") != 0) return 5;
    printf("%s\n", pixo_hip_version());
    return 0;
}
''')
    exe = str(tmp_path / "t")
    lib = os.path.dirname(_lib.__file__)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-o", exe, str(src),
                           "-L" + lib, "-lpixo_hip", "-Wl,-rpath," + lib])
    out = subprocess.run([exe], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert "gfx950" in out


def test_null_arguments_are_errors_not_crashes():
    L = _lib.load()
    o = _lib.JpegOptionsC()
    L.pixo_jpeg_options_from_preset(C.byref(o), 8, 8, 80, 0)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    px = synth.noise(8, 8)
    assert L.pixo_hip_jpeg_encode(px.ctypes.data, px.size, None, C.byref(out), C.byref(n)) == -6
    assert b"null argument 'options'" in L.pixo_hip_last_error()
    assert L.pixo_hip_jpeg_encode(px.ctypes.data, px.size, C.byref(o), None, C.byref(n)) == -6
    assert L.pixo_hip_jpeg_encode_device(None, None, C.byref(out), C.byref(n)) == -6
    L.pixo_hip_coeff_geometry.argtypes = [C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8, C.c_void_p, C.c_void_p]
    assert L.pixo_hip_coeff_geometry(8, 8, 2, 1, None, None) == -6


def test_copy_file_copies_exactly_and_large_results_come_back_as_bytes():
    """pixo_hip_copy_file (host code: the library's copy threads + a huge-page hint) and the Python mirror's use of it
    for large results: same bytes as a plain copy, for sizes around the thresholds and for unaligned ends."""
    L = _lib.load()
    rng = np.random.RandomState(7)
    for n in (1, 4097, (2 << 20) - 1, (2 << 20) + 5, (24 << 20) + 3, (40 << 20) + 1):
        src = rng.randint(0, 256, n + 9, dtype=np.uint8)
        dst = np.zeros(n + 9, dtype=np.uint8)
        L.pixo_hip_copy_file(dst.ctypes.data + 3, src.ctypes.data + 5, n)
        assert dst[:3].sum() == 0 and dst[3 + n:].sum() == 0
        assert np.array_equal(dst[3:3 + n], src[5:5 + n])
        b = _lib.file_bytes(L, src.ctypes.data + 5, n)
        assert isinstance(b, bytes) and len(b) == n and b == src[5:5 + n].tobytes()
    L.pixo_hip_copy_file(None, None, 0)  # nothing to do, no crash
