"""The DEVICE entropy-stage block code (pixo_amd/csrc/jpeg_scan_block.h: zig-zag walk, Huffman
symbols, MSB-first packing at a bit offset, 1-padding) compiled for the host and run block by
block — in reverse order, like lanes racing — against the oracle's entropy-coded segment.
Covers on CPU what jpeg_entropy.hip and jpeg_scan_fused.hip do per lane — both forms of the walk: the
reference-shaped one with visitors and the branch-free one; the workgroup scans, the look-back and the
stuffing copy are checked on the GPU (test_gpu_parity.py, whole-file byte identity)."""
import ctypes as C

import numpy as np
import pytest

import emu_lib as E
import oracle_lib as O
import synth


def _scan_segment(jpeg: bytes) -> bytes:
    """Bytes between the SOS header and EOI."""
    assert jpeg[:2] == b"\xff\xd8" and jpeg[-2:] == b"\xff\xd9"
    i = 2
    while True:
        assert jpeg[i] == 0xFF
        marker, seglen = jpeg[i + 1], int.from_bytes(jpeg[i + 2:i + 4], "big")
        i += 2 + seglen
        if marker == 0xDA:
            return jpeg[i:-2]


def _emu_scan(y, cb, cr, w, h, ct, ss, optimize, flat=False):
    L = E.lib()
    fn = L.emu_scan_flat if flat else L.emu_scan
    fn.restype = C.c_long
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_long]
    L.emu_scan_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    tables = np.zeros(536, np.uint32)
    L.emu_scan_tables(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, w, h, ct, ss, int(optimize), tables.ctypes.data)
    mode = 0 if ct == 0 else (2 if ss == 1 else 1)
    n = y.shape[0] + cb.shape[0] + cr.shape[0]
    out = np.zeros(n * 260 + 64, np.uint8)
    got = fn(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, mode, n, tables.ctypes.data, out.ctypes.data, out.size)
    assert got >= 0
    return out[:got].tobytes()


def _check(px, w, h, ct, ss, q, optimize=False):
    y, cb, cr = O.coeffs(px, w, h, ct, ss, q)
    want = _scan_segment(O.encode_from_coeffs(y, cb, cr, O.make_options(w, h, ct, q, ss, optimize_huffman=optimize)))
    assert _emu_scan(y, cb, cr, w, h, ct, ss, optimize) == want
    # the branch-free walkers of the single-pass kernels (block_pack_flat)
    assert _emu_scan(y, cb, cr, w, h, ct, ss, optimize, flat=True) == want


@pytest.mark.parametrize("mode", [(2, 1), (2, 0), (0, 0)])
@pytest.mark.parametrize("q", [1, 35, 80, 100])
def test_device_block_coder_matches_reference_scan(mode, q):
    ct, ss = mode
    w, h = 72, 40
    px = synth.noise_gray(w, h, q) if ct == 0 else synth.noise(w, h, q)
    _check(px, w, h, ct, ss, q)


@pytest.mark.parametrize("optimize", [False, True])
def test_smooth_and_flat_content_long_zero_runs_and_eob(optimize):
    # gradients / flat blocks: ZRL (16-zero runs), EOB right after DC, all-zero AC blocks
    for gen in (synth.gradient_rgb, synth.flat_blocks):
        _check(gen(96, 64), 96, 64, 2, 1, 80, optimize)
        _check(gen(96, 64), 96, 64, 2, 0, 95, optimize)
    _check(synth.constant(33, 17, 128), 33, 17, 2, 1, 50, optimize)
    _check(synth.checkerboard(64, 64, 3), 64, 64, 2, 0, 100, optimize)


def test_optimised_tables_from_device_histogram_visitor():
    px = synth.noise(120, 56, 9)
    _check(px, 120, 56, 2, 1, 60, optimize=True)
    _check(synth.noise_gray(50, 50, 3), 50, 50, 0, 0, 85, optimize=True)


def test_flat_count_walk_gives_the_visitor_histogram():
    """scan_count_kernel's walk (block_count_flat: walk-table slots, ZRL counted run/16 at a time) against the
    reference-shaped walk with CountVisitor: same 536 counters, nothing counted in a slot that stands for no symbol."""
    L = E.lib()
    L.emu_count_compare.restype = C.c_long
    L.emu_count_compare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p]
    cases = [(synth.noise(200, 120, 3), 200, 120, 2, 1, 80), (synth.flat_blocks(48, 48), 48, 48, 2, 1, 100),
             (synth.gradient_rgb(333, 64), 333, 64, 2, 0, 90), (synth.noise_gray(100, 90, 1), 100, 90, 0, 0, 50),
             (synth.extremes(64, 64, 2), 64, 64, 2, 1, 100), (synth.noise(64, 64, 8), 64, 64, 2, 0, 1),
             (synth.checkerboard(64, 64, 3), 64, 64, 2, 0, 100), (synth.constant(33, 17, 128), 33, 17, 2, 1, 50)]
    for px, w, h, ct, ss, q in cases:
        y, cb, cr = O.coeffs(px, w, h, ct, ss, q)
        mode = 0 if ct == 0 else (2 if ss == 1 else 1)
        n = y.shape[0] + cb.shape[0] + cr.shape[0]
        hist = np.zeros(536, np.uint32)
        assert L.emu_count_compare(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, mode, n, hist.ctypes.data) == 0, (w, h, ct, ss, q)
        assert hist.sum() > n  # at least DC + one more symbol per block on average
    # a hand-made block: one coefficient after 62 zeros (three ZRLs), and a DC difference of the largest category
    y = np.zeros((2, 64), np.int16); y[0, 63] = -5; y[1, 0] = 2047; y[1, 17] = 1
    z = np.zeros((0, 64), np.int16)
    hist = np.zeros(536, np.uint32)
    assert L.emu_count_compare(y.ctypes.data, z.ctypes.data, z.ctypes.data, 0, 2, hist.ctypes.data) == 0
    # (natural index 17 is position 8 of the zig-zag scan: a run of 7)
    assert hist[12 + 0xF0] == 3 and hist[12 + 0xE3] == 1 and hist[12 + 0x71] == 1 and hist[12] == 1 and hist[11] == 1 and hist[0] == 1
    assert hist.sum() == 8


@pytest.mark.parametrize("per_piece", [1, 5, 192, 1000])
def test_a_scan_coded_in_pieces_is_the_same_scan(per_piece):
    """Large scans are coded in pieces that hand each other the bit position (pieces.cpp device_entropy_pieces, pixo_dev::ScanPiece):
    a piece's stream starts with the (bits before) % 8 last bits of the piece before, every piece but the last is stuffed in
    whole bytes only.  The same hand-off on the CPU with the device's flat walk, pieces of 1 block to 1000, against the oracle's scan."""
    L = E.lib()
    L.emu_scan_pieces.restype = C.c_long
    L.emu_scan_pieces.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_long]
    L.emu_scan_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    cases = [(synth.noise(200, 120, 3), 200, 120, 2, 1, 80), (synth.flat_blocks(48, 48), 48, 48, 2, 1, 100),
             (synth.gradient_rgb(333, 64), 333, 64, 2, 0, 90), (synth.noise_gray(100, 90, 1), 100, 90, 0, 0, 50),
             (synth.extremes(64, 64, 2), 64, 64, 2, 1, 100), (synth.constant(33, 17, 128), 33, 17, 2, 1, 50)]
    for px, w, h, ct, ss, q in cases:
        y, cb, cr = O.coeffs(px, w, h, ct, ss, q)
        tables = np.zeros(536, np.uint32)
        L.emu_scan_tables(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, w, h, ct, ss, 0, tables.ctypes.data)
        mode = 0 if ct == 0 else (2 if ss == 1 else 1)
        n = y.shape[0] + cb.shape[0] + cr.shape[0]
        out = np.zeros(n * 260 + 64, np.uint8)
        got = L.emu_scan_pieces(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, mode, n, tables.ctypes.data, per_piece, out.ctypes.data, out.size)
        want = _scan_segment(O.encode_from_coeffs(y, cb, cr, O.make_options(w, h, ct, q, ss)))
        assert got >= 0 and out[:got].tobytes() == want, (w, h, ct, ss, q, per_piece)


def test_ff_bytes_are_stuffed_and_last_byte_padded_with_ones():
    # noise at q=100 produces plenty of 0xFF bytes in the packed stream
    px = synth.noise(64, 64, 4)
    y, cb, cr = O.coeffs(px, 64, 64, 2, 0, 100)
    seg = _emu_scan(y, cb, cr, 64, 64, 2, 0, False)
    assert b"\xff\x00" in seg
    assert all(seg[i + 1] == 0 for i in range(len(seg) - 1) if seg[i] == 0xFF)
    _check(px, 64, 64, 2, 0, 100)


@pytest.mark.parametrize("scratch,window", [(12, 1024), (16, 768), (4, 64), (40, 97)])
def test_single_walk_scratch_and_gather_give_the_same_bits(scratch, window):
    """jpeg_scan_fused.hip codes every block ONCE into a per-lane scratch and gathers the scratches, shifted, into the group's
    bit buffer (window by window); groups that hold a block longer than the scratch take a second walk.  The same
    choreography on the CPU, with small scratches and windows to hit every seam, against the plain packer."""
    L = E.lib()
    L.emu_scan_single_walk.restype = C.c_long
    L.emu_scan_single_walk.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32,
                                       C.c_void_p, C.c_void_p]
    L.emu_scan_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    cases = [(synth.noise(200, 120, 3), 200, 120, 2, 1, 80), (synth.flat_blocks(48, 48), 48, 48, 2, 1, 100),
             (synth.gradient_rgb(333, 64), 333, 64, 2, 0, 90), (synth.noise_gray(100, 90, 1), 100, 90, 0, 0, 50),
             (synth.extremes(64, 64, 2), 64, 64, 2, 1, 100)]
    for px, w, h, ct, ss, q in cases:
        y, cb, cr = O.coeffs(px, w, h, ct, ss, q)
        tables = np.zeros(536, np.uint32)
        L.emu_scan_tables(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, w, h, ct, ss, 0, tables.ctypes.data)
        mode = 0 if ct == 0 else (2 if ss == 1 else 1)
        n = y.shape[0] + cb.shape[0] + cr.shape[0]
        words = np.zeros(n * 60 + 16, np.uint32)
        total = C.c_uint64()
        L.emu_scan_single_walk(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, mode, n, tables.ctypes.data, scratch, window,
                               words.ctypes.data, C.byref(total))
        # the same bits, padded and stuffed by hand, must be the oracle's scan
        bits = total.value
        nbytes = (bits + 7) // 8
        raw = bytearray(words[: (bits + 31) // 32].astype(">u4").tobytes()[:nbytes])
        if bits % 8:
            raw[-1] |= (1 << (8 - bits % 8)) - 1
        stuffed = bytes(raw).replace(b"\xff", b"\xff\x00")
        want = _scan_segment(O.encode_from_coeffs(y, cb, cr, O.make_options(w, h, ct, q, ss)))
        assert stuffed == want, (w, h, ct, ss, q, scratch, window)
