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


def test_end_of_band_runs_longer_than_32767_blocks():
    """A flat image has no AC coefficient anywhere: the AC scans are pure end-of-band runs, flushed
    every 0x7FFF blocks (progressive.rs:186-190).  2048x2048 4:4:4 = 65536 blocks per component (two
    flushes + a tail); a lone textured block in the middle splits one run."""
    w = h = 2048
    px = synth.constant(w, h, 97).reshape(h, w, 3).copy()
    for variant in range(2):
        if variant:
            px[1000:1008, 1200:1208] = synth.noise(8, 8, 3).reshape(8, 8, 3)
        flat = px.reshape(-1)
        for ss in (0, 1):
            got = jpeg.encode(flat, _opts(w, h, 2, ss, 85, progressive=True))
            want = O.encode(flat, O.make_options(w, h, 2, 85, ss, progressive=True))
            assert got == want
    g = px[:, :, 0].reshape(-1).copy()
    assert jpeg.encode(g, _opts(w, h, 0, 0, 85, progressive=True, optimize_huffman=True)) == \
        O.encode(g, O.make_options(w, h, 0, 85, 0, progressive=True, optimize_huffman=True))


@pytest.mark.parametrize("mode", [(2, 1), (2, 0), (0, 0)])
def test_progressive_scans_over_a_device_tuple_and_the_host_twin(mode):
    """`entropy_encode_device` with progressive options codes the seven scans on the device from a tuple
    that never leaves HBM; the host twin (debug switch host_entropy) must give the same bytes."""
    import torch
    ct, ss = mode
    w, h = 413, 290
    px = synth.noise(w, h, 5) if ct == 2 else synth.noise_gray(w, h, 5)
    y, cb, cr = O.coeffs(px, w, h, ct, ss, 88)
    dy = torch.from_numpy(y).to("cuda:0")
    dcb = torch.from_numpy(cb).to("cuda:0") if cb.size else dy
    dcr = torch.from_numpy(cr).to("cuda:0") if cr.size else dy
    torch.cuda.synchronize()
    for optimize in (False, True):
        o = _opts(w, h, ct, ss, 88, progressive=True, optimize_huffman=optimize)
        want = O.encode_from_coeffs(y, cb, cr, O.make_options(w, h, ct, 88, ss, progressive=True, optimize_huffman=optimize)) \
            if not optimize else O.encode(px, O.make_options(w, h, ct, 88, ss, progressive=True, optimize_huffman=True))
        got = jpeg.entropy_encode_device(dy, dcb, dcr, o)
        assert got == want
        jpeg.debug_configure("host_entropy")
        try:
            assert jpeg.entropy_encode_device(dy, dcb, dcr, o) == want
            assert jpeg.encode(px, o) == want
        finally:
            jpeg.debug_configure(None)
        assert jpeg.encode(px, o) == want


def test_progressive_large_noise_matches_host_twin():
    """4096x4096 noise (the stream has millions of 0xFF bytes and every scan is many tiles long): device
    scan coder == host twin byte for byte (the twin is pinned to the oracle on the CPU suite)."""
    w = h = 4096
    px = synth.noise(w, h, 77)
    o = jpeg.JpegOptions.from_preset(w, h, 85, 2)
    dev = jpeg.encode(px, o)
    jpeg.debug_configure("host_entropy")
    try:
        host = jpeg.encode(px, o)
    finally:
        jpeg.debug_configure(None)
    assert hashlib.sha256(dev).hexdigest() == hashlib.sha256(host).hexdigest() and len(dev) == len(host)


def test_random_sweep_of_small_progressive_images():
    """150 random shapes (1x1 upwards: scans of a single block, empty chroma scans, ragged edges),
    qualities, contents and flag combinations against the oracle."""
    rng = np.random.RandomState(2024)
    for i in range(150):
        w, h = int(rng.randint(1, 97)), int(rng.randint(1, 97))
        ct = 2 if rng.rand() < 0.7 else 0
        ss = int(rng.rand() < 0.5)
        q = int(rng.choice([1, 10, 35, 50, 75, 90, 100]))
        kind = i % 4
        if kind == 0:
            rgb = synth.noise(w, h, i)
        elif kind == 1:
            rgb = synth.gradient_rgb(w, h)
        elif kind == 2:
            rgb = synth.constant(w, h, int(rng.randint(0, 256)))
        else:
            rgb = synth.extremes(w, h, i)
        px = rgb if ct == 2 else rgb.reshape(-1, 3)[:, i % 3].copy()
        trellis, optimize = bool(rng.rand() < 0.5), bool(rng.rand() < 0.5)
        got = jpeg.encode(px, _opts(w, h, ct, ss, q, progressive=True, trellis_quant=trellis, optimize_huffman=optimize))
        want = O.encode(px, O.make_options(w, h, ct, q, ss, progressive=True, trellis=trellis, optimize_huffman=optimize))
        assert got == want, (i, w, h, ct, ss, q, kind, trellis, optimize)


def test_progressive_files_straight_into_pinned_caller_storage():
    """`encode_device_into` with progressive options: pinned storage gets the seven scans copied from the device straight
    into it (no pass through the context's buffer), pageable storage one copy from there — same bytes either way, a
    capacity that is one byte short reports the size, and storage with head-room is not touched beyond the file."""
    import torch
    from pixo_amd import error
    for (w, h, ct, ss, kw) in [(1920, 1080, 2, 1, dict(progressive=True)), (333, 77, 2, 0, dict(progressive=True, trellis_quant=True)),
                               (512, 512, 0, 0, dict(progressive=True, optimize_huffman=True, trellis_quant=True))]:
        px = synth.noise(w, h, 77) if ct == 2 else synth.noise_gray(w, h, 77)
        o = _opts(w, h, ct, ss, 85, **kw)
        d = torch.from_numpy(px).to("cuda:0")
        want = jpeg.encode_device(d, o)
        okw = {("trellis" if k == "trellis_quant" else k): v for k, v in kw.items()}
        assert want == O.encode(px, O.make_options(w, h, ct, 85, ss, **okw))
        for arena in (torch.full((len(want) + 40,), 0x5A, dtype=torch.uint8).pin_memory(), torch.full((len(want),), 0x5A, dtype=torch.uint8).pin_memory(),
                      np.full(len(want) + 3, 0x5A, np.uint8)):
            n = jpeg.encode_device_into(arena, d, o)
            raw = arena.numpy() if hasattr(arena, "numpy") else arena
            assert n == len(want) and raw[:n].tobytes() == want
            assert bool((raw[n:] == 0x5A).all())
        with pytest.raises(error.Error, match="need %d bytes" % len(want)):
            jpeg.encode_device_into(torch.zeros(len(want) - 1, dtype=torch.uint8).pin_memory(), d, o)
        # the same from HOST pixels (pixo_hip_jpeg_encode_into): pinned and pageable storage, one byte short
        for arena in (torch.full((len(want) + 9,), 0x5A, dtype=torch.uint8).pin_memory().numpy(), np.full(len(want), 0x5A, np.uint8)):
            n = jpeg.encode_into_buffer(arena, px, o)
            assert n == len(want) and arena[:n].tobytes() == want and bool((arena[n:] == 0x5A).all())
        with pytest.raises(error.BufferTooSmall) as e:
            jpeg.encode_into_buffer(torch.zeros(len(want) - 1, dtype=torch.uint8).pin_memory().numpy(), px, o)
        assert e.value.needed == len(want)


def test_directed_tuples_for_the_run_counter_on_the_device():
    """tests/band_cases.py through `prog_code_kernel` (device tuple -> progressive file): the end-of-band run counter's events at
    every alignment of wavefronts, groups and the 32767 limit — whole files equal to the oracle's."""
    import torch
    import band_cases
    empty = np.zeros((0, 64), np.int16)
    for nblocks, where in band_cases.cases():
        y, w, h = band_cases.tuple_of(nblocks, where)
        want = O.encode_from_coeffs(y, empty, empty, O.make_options(w, h, 0, 50, 0, progressive=True))
        dy = torch.from_numpy(y).to("cuda:0")
        torch.cuda.synchronize()
        got = jpeg.entropy_encode_device(dy, dy, dy, _opts(w, h, 0, 0, 50, progressive=True))
        assert got == want, (nblocks, where[:6])


def test_small_progressive_files_are_stored_straight_into_a_pinned_destination():
    """Round 5: once a context has seen a small progressive file, the next one's scans are stored by the stuffing kernel at their
    places in the FILE in pinned memory (caller's storage when it is large enough for the prediction, else the context's buffer).
    Same bytes as the oracle's either way; nothing behind the file / the capacity is touched; a destination one byte short
    reports the size needed."""
    import torch
    from pixo_amd import error
    w, h = 320, 200
    d_cases = []
    for seed, (trellis, optimize) in ((3, (False, False)), (4, (True, True))):
        px = synth.noise(w, h, seed)
        o = _opts(w, h, 2, 1, 80, progressive=True, trellis_quant=trellis, optimize_huffman=optimize)
        want = O.encode(px, O.make_options(w, h, 2, 80, 1, progressive=True, trellis=trellis, optimize_huffman=optimize))
        d_cases.append((torch.from_numpy(np.ascontiguousarray(px)).cuda(), o, want))
    for d, o, want in d_cases:
        big = torch.full((4 * len(want) + 8192,), 0xA5, dtype=torch.uint8).pin_memory()
        for _ in range(3):  # (the first call of a context sizes the prediction; from the second on the direct path runs)
            big.fill_(0xA5)
            n = jpeg.encode_device_into(big, d, o)
            assert n == len(want) and big[:n].numpy().tobytes() == want
            assert bool((big[n:] == 0xA5).all()), "bytes behind the file were written"
        exact = torch.full((len(want) + 64,), 0xA5, dtype=torch.uint8).pin_memory()
        n = jpeg.encode_device_into(exact[: len(want)], d, o)
        assert n == len(want) and exact[:n].numpy().tobytes() == want and bool((exact[n:] == 0xA5).all())
        short = torch.full((len(want) + 64,), 0xA5, dtype=torch.uint8).pin_memory()
        with pytest.raises(error.BufferTooSmall) as e:
            jpeg.encode_device_into(short[: len(want) - 1], d, o)
        assert e.value.needed == len(want)
        assert bool((short[len(want) - 1:] == 0xA5).all()), "bytes behind the capacity were written"
        assert bytes(jpeg.encode_device(d, o)) == want  # (into the context's own buffer)


def test_trellis_on_eight_lanes_per_block_equals_one_lane_per_block_and_the_oracle():
    """Round 5: up to 32,768 blocks the trellis search runs with a block's eight survivors on eight lanes (jpeg_trellis.hip,
    trellis_lanes_kernel), above with one lane per block.  Both forms forced on the same images (debug switch trellis_form):
    identical files; the small ones are also the oracle's."""
    cases = [(64, 64, 80, 1, synth.noise(64, 64, 1)), (200, 120, 50, 0, synth.photo(200, 120, 2)), (333, 77, 95, 1, synth.gradient_rgb(333, 77)),
             (96, 96, 100, 0, synth.checkerboard(96, 96, 3)), (128, 128, 1, 1, synth.noise(128, 128, 3)), (17, 9, 75, 1, synth.noise(17, 9, 4)),
             (640, 480, 85, 1, synth.photo(640, 480, 5)), (1024, 768, 60, 1, synth.noise(1024, 768, 6)), (8, 8, 90, 0, synth.constant(8, 8, 200))]
    try:
        for w, h, q, ss, px in cases:
            files = {}
            for form in ("lane", "group"):
                jpeg.debug_configure("trellis_form=" + form)
                files[form] = jpeg.encode(px, _opts(w, h, 2, ss, q, progressive=True, trellis_quant=True, optimize_huffman=True))
            assert files["lane"] == files["group"], (w, h, q, ss)
            if w * h <= 333 * 120:
                assert files["group"] == O.encode(px, O.make_options(w, h, 2, q, ss, progressive=True, trellis=True, optimize_huffman=True)), (w, h, q, ss)
    finally:
        jpeg.debug_configure(None)
