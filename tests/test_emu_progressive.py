"""The DEVICE progressive scan coder compiled for the host and run lane by lane against the seven entropy-coded segments
of the oracle's progressive file (standard tables) — in both of its forms: the multi-pass one (jpeg_scan_block.h:
band_flags / band_run_before / prog_emit; blocks packed in reverse order) and the single-pass one of round 4
(band_pack_flat / dc_pack_flat for a lane's own symbols, band_count_in_wave / band_wave_summary / band_edge for what the
end-of-band run counter adds around them, wavefront by wavefront and group by group)."""
import ctypes as C

import numpy as np
import pytest

import emu_lib as E
import oracle_lib as O
import synth


def _segments(jpeg: bytes):
    """The entropy-coded bytes that follow each of the SOS headers."""
    segs, i = [], 2
    while i < len(jpeg):
        assert jpeg[i] == 0xFF
        marker = jpeg[i + 1]
        if marker == 0xD9:
            break
        seglen = int.from_bytes(jpeg[i + 2:i + 4], "big")
        i += 2 + seglen
        if marker == 0xDA:
            j = i
            while not (jpeg[j] == 0xFF and jpeg[j + 1] != 0x00):
                j += 1
            segs.append(jpeg[i:j])
            i = j
    return segs


def _emu(y, cb, cr, entry="emu_progressive"):
    L = E.lib()
    fn = getattr(L, entry)
    fn.restype = C.c_long
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
    L.emu_progressive_tables.argtypes = [C.c_void_p]
    tables = np.zeros(536, np.uint32)
    L.emu_progressive_tables(tables.ctypes.data)
    n = y.shape[0] + 2 * cb.shape[0]
    out = np.zeros(n * 600 + 64, np.uint8)
    seg_len = (C.c_long * 7)()
    total = fn(y.ctypes.data, cb.ctypes.data if cb.size else None, cr.ctypes.data if cr.size else None,
                              y.shape[0], cb.shape[0], tables.ctypes.data, out.ctypes.data, out.size, seg_len)
    assert total >= 0
    segs, o = [], 0
    for i in range(7):
        segs.append(out[o:o + seg_len[i]].tobytes())
        o += seg_len[i]
    return segs


def _check(px, w, h, ct, ss, q):
    y, cb, cr = O.coeffs(px, w, h, ct, ss, q)
    want = _segments(O.encode_from_coeffs(y, cb, cr, O.make_options(w, h, ct, q, ss, progressive=True)))
    assert len(want) == 7
    assert _emu(y, cb, cr) == want
    assert _emu(y, cb, cr, "emu_progressive_flat") == want  # the single-pass form (round 4)
    assert _emu(y, cb, cr, "emu_progressive_one_walk") == want  # one walk per block for all scans of its component (round 5)


@pytest.mark.parametrize("mode", [(2, 1), (2, 0), (0, 0)])
@pytest.mark.parametrize("q", [5, 50, 90, 100])
def test_noise(mode, q):
    ct, ss = mode
    w, h = 88, 56
    _check(synth.noise_gray(w, h, q) if ct == 0 else synth.noise(w, h, q), w, h, ct, ss, q)


def test_smooth_and_flat_content_long_end_of_band_runs():
    for gen, q in ((synth.gradient_rgb, 80), (synth.flat_blocks, 60), (synth.gradient_rgb, 20)):
        _check(gen(256, 96), 256, 96, 2, 1, q)
        _check(gen(200, 64), 200, 64, 2, 0, q)
    _check(synth.constant(128, 128, 77), 128, 128, 2, 1, 75)      # every AC band empty: one long run per scan
    _check(synth.checkerboard(96, 96, 8), 96, 96, 2, 0, 85)


def test_end_of_band_run_reaches_32767():
    # 3 x 32767 + 5 empty blocks in a row: the counter is flushed at 32767 three times
    w, h = 8 * 1024, 8 * 97  # 99,328 gray blocks of a constant image
    _check(synth.constant(w, h, 10, 1), w, h, 0, 0, 50)


def test_synthetic_tuples_run_counter_at_every_alignment_of_groups_wavefronts_and_32767():
    """The end-of-band run counter of the single-pass coder travels through wavefront ballots, three LDS words per group and a
    look-back across groups.  Directed tuples (tests/band_cases.py) put its events at every alignment; both forms of the
    device coder, run on the CPU, against the oracle's segments."""
    import band_cases
    empty = np.zeros((0, 64), np.int16)
    for nblocks, where in band_cases.cases():
        y, w, h = band_cases.tuple_of(nblocks, where)
        want = _segments(O.encode_from_coeffs(y, empty, empty, O.make_options(w, h, 0, 50, 0, progressive=True)))
        assert _emu(y, empty, empty, "emu_progressive_flat") == want, (nblocks, where[:6])
        assert _emu(y, empty, empty, "emu_progressive_one_walk") == want, (nblocks, where[:6])
        assert _emu(y, empty, empty) == want, (nblocks, where[:6])
