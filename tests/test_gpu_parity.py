"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI
(pixo_amd/libpixo_hip.so); the oracle and the reference-made goldens are only the checkers.
Bit-exact: coefficients are integers, files are bytes."""
import hashlib

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
import synth
from pixo_amd import ColorType, jpeg

pytestmark = pytest.mark.gpu


def _opts(w, h, ct, ss, q, **kw):
    b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss))
    for k, v in kw.items():
        b = getattr(b, k)(v)
    return b.build()


def _check_coeffs(px, w, h, ct, ss, q, threads=8):
    oy, ocb, ocr = O.coeffs(px, w, h, ct, ss, q, threads=threads)
    gy, gcb, gcr = jpeg.coefficients(px, _opts(w, h, ct, ss, q))
    assert np.array_equal(gy, oy), "Y coefficients differ"
    assert np.array_equal(gcb, ocb) and np.array_equal(gcr, ocr), "chroma coefficients differ"


def test_native_library_is_the_one_running():
    assert jpeg.device_count() >= 1
    import os
    maps = open("/proc/self/maps").read()
    assert "libpixo_hip.so" in maps


ALL = G.cases(max_pixels=1100 * 1100)


@pytest.mark.parametrize("c", ALL, ids=[c["name"] for c in ALL])
def test_whole_file_bytes_match_reference_goldens(c):
    """encode_jpeg() — the reference's flat entry — must return the reference's bytes."""
    blob = jpeg.encode_jpeg(G.make_input(c), c["w"], c["h"], c["color_type"], c["quality"], c["preset"], c["s420"])
    G.check(c, blob)


@pytest.mark.parametrize("w,h", [(1, 1), (2, 2), (7, 7), (8, 8), (9, 9), (16, 16), (15, 17), (1, 100), (100, 1),
                                 (256, 256), (512, 512), (1000, 1000), (1024, 1024), (511, 16), (513, 17),
                                 (1028, 33), (2048, 16), (1918, 70), (1921, 40)])
@pytest.mark.parametrize("mode", [(2, 1), (2, 0), (0, 0)])
def test_coefficients_edge_dimensions(w, h, mode):
    """reference EDGE_CASE_DIMENSIONS (tests/support/synthetic.rs:276) + tile-boundary sizes."""
    ct, ss = mode
    px = synth.noise_gray(w, h, 17) if ct == 0 else synth.noise(w, h, 17)
    _check_coeffs(px, w, h, ct, ss, 80)


@pytest.mark.parametrize("q", [1, 2, 10, 35, 49, 50, 51, 75, 80, 85, 95, 99, 100])
def test_coefficients_quality_sweep(q):
    for gen in (synth.noise(200, 120, q), synth.gradient_rgb(200, 120), synth.flat_blocks(200, 120),
                synth.checkerboard(200, 120, 5)):
        _check_coeffs(gen, 200, 120, 2, 1, q)
        _check_coeffs(gen, 200, 120, 2, 0, q)


def test_saturated_colours():
    for rgb in ([0, 0, 255], [255, 0, 0], [255, 255, 255], [0, 0, 0], [0, 255, 0], [1, 0, 255]):
        px = np.tile(np.array(rgb, np.uint8), 64 * 32)
        _check_coeffs(px, 64, 32, 2, 1, 100)
        _check_coeffs(px, 64, 32, 2, 0, 100)
    for seed in (1, 2, 3):  # clamped and unclamped pixels inside the same 2x2 box / block
        _check_coeffs(synth.extremes(600, 48, seed), 600, 48, 2, 1, 100)
        _check_coeffs(synth.extremes(600, 48, seed), 600, 48, 2, 0, 90)


def test_config1_512_and_config3_unit_1080p_files():
    for (w, h) in [(512, 512), (1920, 1080)]:
        c = [c for c in G.load()["cases"] if (c["w"], c["h"], c["preset"], c["gen"]) == (w, h, 0, "noise")][0]
        G.check(c, jpeg.encode_jpeg(G.make_input(c), w, h, 2, 80, 0, True))


def test_config2_4096_coefficients_and_file_hashes():
    """BASELINE config 2: 4096x4096 q80, both subsamplings; coefficient tuple vs the oracle and
    whole-file sha256 vs the reference-made golden."""
    w = h = 4096
    px = synth.noise(w, h, 42)
    for ss, s420 in ((1, True), (0, False)):
        _check_coeffs(px, w, h, 2, ss, 80)
        c = [c for c in G.load()["cases"] if (c["w"], c["s420"], c["gen"]) == (4096, s420, "noise")][0]
        G.check(c, jpeg.encode(px, _opts(w, h, 2, ss, 80)))


def test_restart_and_optimized_huffman_whole_path():
    w, h = 333, 211
    px = synth.noise(w, h, 5)
    for ss in (0, 1):
        for restart, opt in [(None, True), (5, False), (64, True)]:
            got = jpeg.encode(px, _opts(w, h, 2, ss, 66, restart_interval=restart, optimize_huffman=opt))
            want = O.encode(px, O.make_options(w, h, 2, 66, ss, restart=restart, optimize_huffman=opt))
            assert got == want


def test_device_api_batch_of_1080p_matches_per_image_oracle():
    """BASELINE config 3 shape: a batch of 1920x1080 images in one launch on device memory
    (8 here to keep the CPU oracle quick; bench covers 64)."""
    import torch
    w, h, n = 1920, 1080, 8
    imgs = [synth.noise(w, h, 42 + i) for i in range(n)]
    yb, cbn = jpeg.coefficient_geometry(w, h, 2, 1)
    dev = torch.device("cuda:0")
    d_px = torch.from_numpy(np.concatenate(imgs)).to(dev)
    d_y = torch.empty((n * yb, 64), dtype=torch.int16, device=dev)
    d_cb = torch.empty((n * cbn, 64), dtype=torch.int16, device=dev)
    d_cr = torch.empty((n * cbn, 64), dtype=torch.int16, device=dev)
    jpeg.coefficients_device(d_px, w, h, 2, 1, 80, d_y, d_cb, d_cr, batch=n,
                             stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    y, cb, cr = d_y.cpu().numpy(), d_cb.cpu().numpy(), d_cr.cpu().numpy()
    for i in range(n):  # every image of the batch (round 3 compared three of the eight)
        oy, ocb, ocr = O.coeffs(imgs[i], w, h, 2, 1, 80, threads=8)
        assert np.array_equal(y[i * yb:(i + 1) * yb], oy)
        assert np.array_equal(cb[i * cbn:(i + 1) * cbn], ocb) and np.array_equal(cr[i * cbn:(i + 1) * cbn], ocr)
    # image 0 inside the batch == image 0 alone (launch-shape independence)
    gy, gcb, gcr = jpeg.coefficients(imgs[0], _opts(w, h, 2, 1, 80))
    assert np.array_equal(gy, y[:yb]) and np.array_equal(gcb, cb[:cbn])


def test_bands_concatenate_to_the_single_device_tuple():
    """Config 4's sharding rule at a size one GPU and the CPU oracle finish quickly: 8 MCU-row
    bands computed independently (as 8 GPUs would) == the whole-image tuple, and the host
    entropy stage over the stitched tuple gives the oracle's file."""
    w, h = 2048, 1000
    px = synth.noise(w, h, 8)
    full = jpeg.coefficients(px, _opts(w, h, 2, 1, 80))
    parts = 8
    ys, cbs, crs = [], [], []
    for i in range(parts):
        b = jpeg.band(w, h, 2, 1, parts, i)
        sub = px[b["row_begin"] * w * 3: b["row_end"] * w * 3]
        y, cb, cr = jpeg.coefficients(sub, _opts(w, b["row_end"] - b["row_begin"], 2, 1, 80))
        assert y.shape[0] == b["y_blocks"]
        ys.append(y); cbs.append(cb); crs.append(cr)
    y, cb, cr = np.concatenate(ys), np.concatenate(cbs), np.concatenate(crs)
    assert np.array_equal(y, full[0]) and np.array_equal(cb, full[1]) and np.array_equal(cr, full[2])
    assert jpeg.entropy_encode(y, cb, cr, _opts(w, h, 2, 1, 80)) == O.encode(px, O.make_options(w, h, 2, 80, 1))


def test_determinism_and_linearity_properties_at_full_size():
    """Size-independent properties on a 4096x4096 image the oracle is not consulted for:
    (i) two runs are identical; (ii) vertical flip of a 16-row-aligned image permutes MCU rows
    of |DC| identically (DC is the block mean: flipping rows inside a block keeps it)."""
    w = h = 4096
    px = synth.noise(w, h, 123)
    o = _opts(w, h, 2, 1, 80)
    a = jpeg.coefficients(px, o)
    b = jpeg.coefficients(px, o)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    flipped = px.reshape(h, w * 3)[::-1].reshape(-1)
    f = jpeg.coefficients(np.ascontiguousarray(flipped), o)
    mw = w // 16
    cb = a[1][:, 0].reshape(-1, mw)
    cbf = f[1][:, 0].reshape(-1, mw)[::-1]
    assert np.array_equal(cb, cbf)  # chroma DC of an MCU is invariant under the flip


# ---- device entropy stage (jpeg_entropy.hip) ----------------------------------------------------

def _file_from_device_tuple(px, w, h, ct, ss, q, **kw):
    """pixels -> device tuple (coefficient kernel) -> device entropy stage -> file bytes"""
    import torch
    dev = torch.device("cuda:0")
    yb, cbn = jpeg.coefficient_geometry(w, h, ct, ss)
    d_px = torch.from_numpy(np.ascontiguousarray(px, np.uint8)).to(dev)
    d_y = torch.empty((yb, 64), dtype=torch.int16, device=dev)
    d_cb = torch.empty((max(cbn, 1), 64), dtype=torch.int16, device=dev)
    d_cr = torch.empty((max(cbn, 1), 64), dtype=torch.int16, device=dev)
    jpeg.coefficients_device(d_px, w, h, ct, ss, q, d_y, d_cb, d_cr, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return jpeg.entropy_encode_device(d_y, d_cb, d_cr, _opts(w, h, ct, ss, q, **kw))


@pytest.mark.parametrize("mode", [(2, 1), (2, 0), (0, 0)])
@pytest.mark.parametrize("optimize", [False, True])
def test_device_entropy_stage_files_equal_the_oracle(mode, optimize):
    ct, ss = mode
    for (w, h, q, gen) in [(200, 120, 80, "noise"), (333, 77, 35, "noise"), (640, 480, 95, "gradient"),
                           (96, 96, 100, "noise"), (1024, 64, 1, "noise"), (17, 9, 60, "noise")]:
        if gen == "gradient":
            px = synth.gradient_rgb(w, h) if ct == 2 else synth.gradient_rgb(w, h).reshape(-1, 3)[:, 1].copy()
        else:
            px = synth.noise(w, h, q) if ct == 2 else synth.noise_gray(w, h, q)
        got = _file_from_device_tuple(px, w, h, ct, ss, q, optimize_huffman=optimize)
        want = O.encode(px, O.make_options(w, h, ct, q, ss, optimize_huffman=optimize))
        assert got == want, (w, h, q, gen)


def test_device_entropy_stage_long_zero_runs_flat_and_tiny_images():
    for px, w, h in [(synth.constant(64, 64, 128), 64, 64), (synth.flat_blocks(160, 96), 160, 96),
                     (synth.checkerboard(128, 128, 16), 128, 128), (synth.constant(1, 1, 7), 1, 1),
                     (synth.extremes(80, 48, 2), 80, 48)]:
        for ss in (0, 1):
            assert _file_from_device_tuple(px, w, h, 2, ss, 75) == O.encode(px, O.make_options(w, h, 2, 75, ss))


@pytest.mark.parametrize("mode", [(2, 1), (2, 0), (0, 0)])
def test_device_entropy_stage_restart_markers(mode):
    """RSTn: byte-aligned segments with 1-padding, predictors reset, FF D0..D7 cycling, no marker
    after the last segment (jpeg/mod.rs:1423-1445) — all on the device."""
    ct, ss = mode
    w, h = 300, 200
    px = synth.noise(w, h, 11) if ct == 2 else synth.noise_gray(w, h, 11)
    for restart, opt in [(1, False), (2, True), (7, False), (19, True), (64, False), (10_000, False)]:
        # 10_000 > MCU count: DRI header but no marker
        got = _file_from_device_tuple(px, w, h, ct, ss, 70, restart_interval=restart, optimize_huffman=opt)
        assert got == O.encode(px, O.make_options(w, h, ct, 70, ss, restart=restart, optimize_huffman=opt)), (restart, opt)
    # smooth content: many 0xFF-free, short segments; and an interval that divides the MCU count exactly
    g = synth.gradient_rgb(256, 64) if ct == 2 else synth.gradient_rgb(256, 64).reshape(-1, 3)[:, 0].copy()
    for restart in (4, 16, 32):
        assert _file_from_device_tuple(g, 256, 64, ct, ss, 90, restart_interval=restart) == \
            O.encode(g, O.make_options(256, 64, ct, 90, ss, restart=restart))


def test_restart_markers_at_full_size_whole_path():
    w = h = 2048
    px = synth.noise(w, h, 21)
    for restart in (1, 128):
        got = jpeg.encode(px, _opts(w, h, 2, 1, 80, restart_interval=restart))
        assert got == O.encode(px, O.make_options(w, h, 2, 80, 1, restart=restart))


def test_encode_device_resident_pixels_full_size_hashes():
    """configs[1] through encode_device(): the reference's own 4096x4096 files."""
    import torch
    w = h = 4096
    d_px = torch.from_numpy(synth.noise(w, h, 42)).to("cuda:0")
    torch.cuda.synchronize()
    for ss, n, sha in [(1, 11150133, "0e8ec9217215507f1ac235c22a969586906a267370249aa8653adcad5e472e49"),
                       (0, 23336677, "831e247fe60d1aa4bb9b43e83f5744f9b061c77f05017f67830f339c70a0a564")]:
        blob = jpeg.encode_device(d_px, _opts(w, h, 2, ss, 80))
        assert len(blob) == n and hashlib.sha256(blob).hexdigest() == sha
    # repeated calls reuse the context's buffers
    assert jpeg.encode_device(d_px, _opts(w, h, 2, 1, 80)) == jpeg.encode_device(d_px, _opts(w, h, 2, 1, 80))


def test_banded_encode_over_rccl_world_of_one():
    """sharded.encode_banded on the real device path (RCCL process group, device tensors, device band
    encoder) and the gathered fallback.  Only one GPU is available to the tests, so the world has one rank;
    the multi-rank exchanges run on CPU over gloo (tests/test_sharding_gloo.py) and, inside one process,
    in tests/test_gpu_multi.py."""
    import os
    import torch
    import torch.distributed as dist
    from pixo_amd import sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        for (w, h, ct, ss) in [(1024, 520, 2, 1), (300, 100, 2, 0), (200, 64, 0, 0)]:
            px = synth.noise(w, h, 31) if ct == 2 else synth.noise_gray(w, h, 31)
            d_px = torch.from_numpy(px).to("cuda:0")
            want = O.encode(px, O.make_options(w, h, ct, 80, ss))
            assert sharded.encode_banded(d_px, _opts(w, h, ct, ss, 80)) == want      # device pixels
            assert sharded.encode_banded(px, _opts(w, h, ct, ss, 80)) == want        # host pixels
            assert sharded.encode_banded(d_px, _opts(w, h, ct, ss, 80, optimize_huffman=True)) == \
                O.encode(px, O.make_options(w, h, ct, 80, ss, optimize_huffman=True))
            got = sharded.encode_gathered_device(d_px, _opts(w, h, ct, ss, 80, restart_interval=7))
            assert got == O.encode(px, O.make_options(w, h, ct, 80, ss, restart=7))
    finally:
        dist.destroy_process_group()


def test_random_option_sweep_whole_files():
    """40 pseudo-random (size, mode, quality, restart, tables, content) combinations through
    encode(): every file byte-identical to the oracle's."""
    rng = np.random.RandomState(20240917)
    for i in range(40):
        w, h = int(rng.randint(1, 700)), int(rng.randint(1, 300))
        ct, ss = [(2, 1), (2, 0), (0, 0)][rng.randint(3)]
        q = int(rng.choice([1, 10, 37, 50, 75, 80, 92, 100]))
        restart = [None, None, 1, 3, 17, 250][rng.randint(6)]
        opt = bool(rng.randint(2))
        kind = rng.randint(3)
        if ct == 2:
            px = [synth.noise(w, h, i), synth.gradient_rgb(w, h), synth.extremes(w, h, i)][kind]
        else:
            px = [synth.noise_gray(w, h, i), synth.constant(w, h, 200, 1), (synth.noise_gray(w, h, i) >> 5) * 32][kind]
        got = jpeg.encode(px, _opts(w, h, ct, ss, q, restart_interval=restart, optimize_huffman=opt))
        want = O.encode(px, O.make_options(w, h, ct, q, ss, restart=restart, optimize_huffman=opt))
        assert got == want, (i, w, h, ct, ss, q, restart, opt, kind)


@pytest.mark.parametrize("shape", [(1920, 1080, 2, 1, 6), (333, 77, 2, 0, 9), (64, 64, 0, 0, 5), (100, 36, 2, 1, 3)])
def test_batch_whole_files_one_entropy_pass(shape):
    """config 3 shape through encode_batch_device: every image of the batch equals its own oracle file."""
    import torch
    w, h, ct, ss, n = shape
    imgs = [(synth.noise(w, h, 100 + i) if ct == 2 else synth.noise_gray(w, h, 100 + i)) for i in range(n)]
    imgs[1] = synth.gradient_rgb(w, h) if ct == 2 else synth.constant(w, h, 90, 1)  # a very short segment among long ones
    d_px = torch.from_numpy(np.concatenate(imgs)).to("cuda:0")
    torch.cuda.synchronize()
    files = jpeg.encode_batch_device(d_px, _opts(w, h, ct, ss, 80), n)
    assert len(files) == n
    for i in range(n):
        assert files[i] == O.encode(imgs[i], O.make_options(w, h, ct, 80, ss)), i
    # options that need per-image work take the one-by-one route and still agree
    files = jpeg.encode_batch_device(d_px, _opts(w, h, ct, ss, 80, optimize_huffman=True, restart_interval=5), n)
    for i in (0, n - 1):
        assert files[i] == O.encode(imgs[i], O.make_options(w, h, ct, 80, ss, optimize_huffman=True, restart=5)), i


@pytest.mark.parametrize("shape", [(1920, 1080, 2, 1, 5), (333, 77, 2, 0, 9), (64, 64, 0, 0, 5), (16, 16, 2, 1, 40)])
def test_batch_into_one_arena_every_file_at_its_final_place(shape):
    """`pixo_hip_jpeg_encode_batch_device_into`: the files back to back in caller storage (pinned and pageable), offsets and
    lengths reported, each file equal to its own oracle file; the size query; a capacity that is one byte short."""
    import torch
    from pixo_amd import error
    w, h, ct, ss, n = shape
    imgs = [(synth.noise(w, h, 300 + i) if ct == 2 else synth.noise_gray(w, h, 300 + i)) for i in range(n)]
    imgs[n // 2] = synth.gradient_rgb(w, h) if ct == 2 else synth.constant(w, h, 31, 1)
    want = [O.encode(im, O.make_options(w, h, ct, 80, ss)) for im in imgs]
    d_px = torch.from_numpy(np.concatenate(imgs)).to("cuda:0")
    torch.cuda.synchronize()
    o = _opts(w, h, ct, ss, 80)
    offs, lens = jpeg.encode_batch_device_into(None, d_px, o, n)  # size query
    assert lens == [len(f) for f in want] and offs == [sum(lens[:i]) for i in range(n)]
    total = sum(lens)
    for arena in (torch.full((total + 64,), 0x5A, dtype=torch.uint8).pin_memory(), torch.full((total,), 0x5A, dtype=torch.uint8),
                  np.full(total + 7, 0x5A, np.uint8)):
        offs, lens = jpeg.encode_batch_device_into(arena, d_px, o, n)
        raw = arena.numpy() if hasattr(arena, "numpy") else arena
        for i in range(n):
            assert raw[offs[i]: offs[i] + lens[i]].tobytes() == want[i], i
        assert bool((raw[total:] == 0x5A).all())
    small = torch.zeros(total - 1, dtype=torch.uint8).pin_memory()
    with pytest.raises(error.Error, match="need %d bytes" % total):
        jpeg.encode_batch_device_into(small, d_px, o, n)
    # per-image option sets go one by one, into the same layout
    o2 = _opts(w, h, ct, ss, 80, optimize_huffman=True)
    want2 = [O.encode(im, O.make_options(w, h, ct, 80, ss, optimize_huffman=True)) for im in imgs]
    arena = torch.zeros(sum(len(f) for f in want2) + 16, dtype=torch.uint8).pin_memory()
    offs, lens = jpeg.encode_batch_device_into(arena, d_px, o2, n)
    for i in range(n):
        assert arena.numpy()[offs[i]: offs[i] + lens[i]].tobytes() == want2[i], i


def test_restart_intervals_in_the_single_pass_kernels_and_around_their_threshold():
    """Restart intervals of 96 blocks or more (16 MCUs of 4:2:0, 32 of 4:4:4, 96 gray blocks) are segments of the
    single-pass kernels — groups aligned with the segments, RSTn written by the stuffing kernel —, shorter ones take the
    multi-pass kernels: intervals on both sides of the threshold, intervals that divide the image and that do not, a
    last segment of one MCU, an MCU row, images of many tiles per segment; all against the oracle."""
    cases = [(640, 480, 2, 1, (15, 16, 17, 40, 1199, 1200)), (640, 480, 2, 0, (31, 32, 33, 80, 4799)), (512, 384, 0, 0, (95, 96, 97, 3071)),
             (2048, 2048, 2, 1, (128, 1000, 16383)), (4096, 512, 2, 0, (512, 4097)),
             (2048, 2048, 0, 0, (96,))]  # (683 segments: more block sums than one 4 KiB copy holds — sup_layout's second branch)
    for w, h, ct, ss, intervals in cases:
        px = synth.noise(w, h, w + h) if ct == 2 else synth.noise_gray(w, h, w + h)
        smooth = synth.gradient_rgb(w, h) if ct == 2 else synth.gradient_rgb(w, h).reshape(-1, 3)[:, 1].copy()
        for r in intervals:
            for img, q in ((px, 80), (smooth, 92)):
                got = jpeg.encode(img, _opts(w, h, ct, ss, q, restart_interval=r))
                assert got == O.encode(img, O.make_options(w, h, ct, q, ss, restart=r)), (w, h, ct, ss, r, q)
    # optimised tables with restart markers (the statistics honour the resets) in the single-pass kernels
    w, h = 800, 608
    px = synth.noise(w, h, 4)
    for r in (16, 50, 1900):
        assert jpeg.encode(px, _opts(w, h, 2, 1, 75, restart_interval=r, optimize_huffman=True)) == \
            O.encode(px, O.make_options(w, h, 2, 75, 1, restart=r, optimize_huffman=True)), r


def test_single_pass_kernels_that_give_up_waiting_fall_back_to_the_multi_pass_kernels():
    """The look-back kernels bound their waits.  With a budget of zero polls every wait for another
    workgroup fails at once: the kernels raise their abort flag instead of spinning, the host sees it in the pinned mailbox
    and codes the scan again with the multi-pass kernels — same bytes, no hang; the fallback counter shows that it happened.
    In a fresh process (the switch is read at start-up) over plain files, pieces, batches, restart segments, bands and
    progressive files."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np, torch; sys.path.insert(0, 'tests'); import synth, oracle_lib as O; from pixo_amd import jpeg\n"
            "def opts(w, h, ss, **kw):\n"
            "    b = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling(ss))\n"
            "    for k, v in kw.items(): b = getattr(b, k)(v)\n"
            "    return b.build()\n"
            "n0 = jpeg.lookback_fallbacks()\n"
            "for (w, h, ss) in ((1024, 768, 1), (777, 333, 0), (4096, 4096, 1)):\n"
            "    px = synth.noise(w, h, 5)\n"
            "    assert jpeg.encode(px, opts(w, h, ss)) == O.encode(px, O.make_options(w, h, 2, 80, ss)), (w, h, ss)\n"
            "    d = torch.from_numpy(px).to('cuda:0'); pin = torch.zeros(w * h * 3, dtype=torch.uint8).pin_memory()\n"
            "    n = jpeg.encode_device_into(pin, d, opts(w, h, ss))\n"
            "    assert pin[:n].numpy().tobytes() == O.encode(px, O.make_options(w, h, 2, 80, ss))\n"
            "n1 = jpeg.lookback_fallbacks(); assert n1 > n0, (n0, n1)\n"
            "px = synth.noise(640, 480, 6)\n"
            "assert jpeg.encode(px, opts(640, 480, 1, restart_interval=40)) == O.encode(px, O.make_options(640, 480, 2, 80, 1, restart=40))\n"
            "imgs = [synth.noise(320, 240, 50 + i) for i in range(6)]\n"
            "files = jpeg.encode_batch_device(torch.from_numpy(np.concatenate(imgs)).to('cuda:0'), opts(320, 240, 1), 6)\n"
            "assert all(files[i] == O.encode(imgs[i], O.make_options(320, 240, 2, 80, 1)) for i in range(6))\n"
            "px = synth.noise(520, 330, 12)\n"
            "assert jpeg.encode_multi(px, opts(520, 330, 1, optimize_huffman=True), [0, 0, 0]) == O.encode(px, O.make_options(520, 330, 2, 80, 1, optimize_huffman=True))\n"
            "n2 = jpeg.lookback_fallbacks(); assert n2 > n1 + 2, (n1, n2)\n"
            # round 4: the single-pass progressive coder (prog_code_kernel, three look-backs) falls back the same way
            "for (w, h, ss, kw) in ((1024, 768, 1, {}), (777, 333, 0, dict(optimize_huffman=True)), (640, 480, 1, dict(trellis_quant=True))):\n"
            "    px = synth.noise(w, h, 7)\n"
            "    okw = {('trellis' if k == 'trellis_quant' else k): v for k, v in kw.items()}\n"
            "    assert jpeg.encode(px, opts(w, h, ss, progressive=True, **kw)) == O.encode(px, O.make_options(w, h, 2, 80, ss, progressive=True, **okw)), (w, h, kw)\n"
            "n3 = jpeg.lookback_fallbacks(); assert n3 >= n2 + 3, (n2, n3)\n"
            "print('fallbacks', n3)")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PIXO_HIP_DEBUG="spin_budget=0"),
                       timeout=600, cwd=root)
    assert r.returncode == 0 and "fallbacks" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    # and in THIS process (default budget) nothing ever falls back
    assert jpeg.lookback_fallbacks() == 0


def test_config4_16384_image_on_one_gpu_matches_the_reference_file():
    """configs[3] at full size on a single MI355X (0.8 GB of pixels, 6.3 M blocks, a 178 MB file):
    64-bit offsets everywhere, and the reference's own result for this input (made by
    the wasm build) — 178,548,465 bytes, sha256 77cc6cb6..."""
    import torch
    w = h = 16384
    px = synth.noise(w, h, 42)
    d_px = torch.from_numpy(px).to("cuda:0")
    del px
    torch.cuda.synchronize()
    blob = jpeg.encode_device(d_px, _opts(w, h, 2, 1, 80))
    assert len(blob) == 178548465
    assert hashlib.sha256(blob).hexdigest() == "77cc6cb69a782693c46f2024ac57ebfdfb8411148fa3cef62698c727af36c70c"


def test_concurrent_calls_from_many_threads_give_the_serial_results():
    """The reference's encode is re-entrant and its users call it from rayon workers: the
    C ABI keeps its context (stream, buffers) per thread.  Twelve threads, each encoding its own
    images with its own options (baseline, optimised tables, restart markers, preset 2, PNG filters),
    repeatedly and at the same time (ctypes drops the GIL): every result equals the one computed
    alone beforehand."""
    import threading
    from pixo_amd import png
    jobs = []
    for t in range(12):
        w, h = 160 + 37 * t, 120 + 29 * (t % 5)
        px = synth.noise(w, h, 100 + t)
        b = jpeg.JpegOptions.builder(w, h).quality(40 + 5 * t).subsampling(jpeg.Subsampling(t & 1))
        if t % 4 == 1: b = b.optimize_huffman(True)
        if t % 4 == 2: b = b.restart_interval(5)
        if t % 4 == 3: b = b.preset(2)
        rgba = synth.lcg_bytes(w * h * 4, 7 + t)
        jobs.append((px, b.build(), rgba, w, h))
    expected = [(jpeg.encode(px, o), png.apply_filters(rgba, w, h, 4, png.FilterStrategy.ADAPTIVE)) for px, o, rgba, w, h in jobs]
    errors = []
    start = threading.Barrier(len(jobs))

    def worker(i):
        px, o, rgba, w, h = jobs[i]
        try:
            start.wait()
            for _ in range(6):
                if jpeg.encode(px, o) != expected[i][0]:
                    errors.append((i, "jpeg"))
                got = png.apply_filters(rgba, w, h, 4, png.FilterStrategy.ADAPTIVE)
                if not (np.array_equal(got[0], expected[i][1][0]) and got[1] == expected[i][1][1]):
                    errors.append((i, "png"))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(jobs))]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors


def test_encode_into_caller_storage_and_the_reserve_and_retry_protocol():
    """pixo_hip_jpeg_encode_into: the file lands in the caller's buffer; a buffer that is too small is
    left untouched, the call reports the needed size, and the retry with that size succeeds.  Every
    flavour of file (baseline, optimised tables, restart markers, progressive + trellis)."""
    from pixo_amd import error
    w, h = 321, 203
    px = synth.noise(w, h, 9)
    B = jpeg.JpegOptions.builder
    for o in (B(w, h).quality(77).subsampling(jpeg.Subsampling.S420).build(),
              B(w, h).quality(60).optimize_huffman(True).build(),
              B(w, h).quality(60).restart_interval(3).build(),
              B(w, h).quality(85).preset(2).build()):
        want = jpeg.encode(px, o)
        small = np.full(len(want) - 1, 0xA5, np.uint8)
        with pytest.raises(error.BufferTooSmall) as ei:
            jpeg.encode_into_buffer(small, px, o)
        assert ei.value.needed == len(want) and (small == 0xA5).all()
        buf = np.zeros(ei.value.needed + 7, np.uint8)
        n = jpeg.encode_into_buffer(buf, px, o)
        assert n == len(want) and buf[:n].tobytes() == want and not buf[n:].any()
    with pytest.raises(error.InvalidQuality):
        jpeg.encode_into_buffer(np.zeros(10, np.uint8), px, B(w, h).quality(0).build())


def test_threads_that_end_give_their_device_buffers_back():
    """A server with one thread per request: 24 short-lived threads in sequence, each encoding a 2048x2048
    image (about 70 MB of per-thread device and pinned buffers).  A thread that ends parks its context in the
    library's pool and the next thread adopts it, so the device's free memory does not shrink by 24 contexts."""
    import threading
    import torch
    w = h = 2048
    px = synth.noise(w, h, 3)
    o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420).build()
    want = jpeg.encode(px, o)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(24):
        res = []
        t = threading.Thread(target=lambda: res.append(jpeg.encode(px, o)))
        t.start(); t.join()
        assert res[0] == want
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 200 << 20, (free0 - free1) >> 20  # 24 leaked contexts would be > 1 GB
    # 40 threads alive at once leave 40 parked contexts behind; the pool keeps 16, the rest is freed by the next
    # thread that asks for one, and pixo_hip_trim frees them all
    start = threading.Barrier(40)
    res = []

    def burst():
        start.wait()
        res.append(jpeg.encode(px, o) == want)

    ts = [threading.Thread(target=burst) for _ in range(40)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert all(res) and len(res) == 40
    t = threading.Thread(target=lambda: res.append(jpeg.encode(px, o) == want))
    t.start(); t.join()
    jpeg.trim()
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info()
    assert free0 - free2 < 200 << 20, (free0 - free2) >> 20


def test_trim_releases_the_threads_buffers_and_the_next_call_starts_over():
    import torch
    w = h = 4096
    px = synth.noise(w, h, 42)
    o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420).build()
    a = jpeg.encode(px, o)
    torch.cuda.synchronize()
    held, _ = torch.cuda.mem_get_info()
    jpeg.trim()
    freed, _ = torch.cuda.mem_get_info()
    assert freed - held > 100 << 20  # pixels + tuple + entropy buffers of a 4096x4096 image
    assert jpeg.encode(px, o) == a
    jpeg.trim(); jpeg.trim()  # idempotent


def test_encode_device_into_pinned_and_pageable_storage():
    """pixo_hip_jpeg_encode_device_into: resident pixels -> caller storage.  Pinned torch tensor and plain
    numpy array; baseline, optimised tables, restart markers, progressive; size query and too-small buffers
    (nothing written)."""
    import torch
    from pixo_amd import error
    w, h = 640, 360
    px = synth.noise(w, h, 21)
    d_px = torch.from_numpy(px).to("cuda:0")
    torch.cuda.synchronize()
    B = jpeg.JpegOptions.builder
    for o in (B(w, h).quality(80).subsampling(jpeg.Subsampling.S420).build(), B(w, h).quality(55).optimize_huffman(True).build(),
              B(w, h).quality(70).restart_interval(4).build(), B(w, h).quality(85).preset(2).build()):
        want = jpeg.encode(px, o)
        pinned = torch.full((len(want) + 64,), 0x5A, dtype=torch.uint8).pin_memory()
        n = jpeg.encode_device_into(pinned, d_px, o)
        assert n == len(want) and pinned[:n].numpy().tobytes() == want and bool((pinned[n:] == 0x5A).all())
        plain = np.zeros(len(want), np.uint8)
        assert jpeg.encode_device_into(plain, d_px, o) == len(want) and plain.tobytes() == want
        small = np.full(len(want) - 1, 0x77, np.uint8)
        with pytest.raises(error.BufferTooSmall) as ei:
            jpeg.encode_device_into(small, d_px, o)
        assert ei.value.needed == len(want) and (small == 0x77).all()


def _pinned_storage_cases():
    """Pinned storage: every alignment of the destination and of the scan's first byte, the bytes around the file
    untouched, a buffer of exactly the file's size, one that is a byte short (size reported)."""
    import torch
    from pixo_amd import error
    B = jpeg.JpegOptions.builder
    for (w, h, q, ss, opt) in ((640, 360, 80, 1, False), (33, 50, 95, 0, False), (1000, 700, 100, 1, False), (512, 512, 30, 1, True),
                               (8, 8, 50, 0, False), (2048, 1024, 90, 1, False)):
        px = synth.noise(w, h, w + q)
        if q == 100:
            px[: len(px) // 2] = 255  # runs of 0xFF bytes in the stream as well
        d_px = torch.from_numpy(px).to("cuda:0")
        torch.cuda.synchronize()
        o = B(w, h).quality(q).subsampling(jpeg.Subsampling(ss)).optimize_huffman(opt).build()
        want = jpeg.encode(px, o)
        for lead in range(5):
            store = torch.full((lead + len(want) + 37,), 0xA5, dtype=torch.uint8).pin_memory()
            n = jpeg.encode_device_into(store[lead:], d_px, o)
            got = store.numpy()
            assert n == len(want) and got[lead:lead + n].tobytes() == want, (w, h, q, lead)
            assert (got[:lead] == 0xA5).all() and (got[lead + n:] == 0xA5).all(), (w, h, q, lead)
        exact = torch.zeros(len(want), dtype=torch.uint8).pin_memory()
        assert jpeg.encode_device_into(exact, d_px, o) == len(want) and exact.numpy().tobytes() == want
        short = torch.zeros(len(want) - 1, dtype=torch.uint8).pin_memory()
        with pytest.raises(error.BufferTooSmall) as ei:
            jpeg.encode_device_into(short, d_px, o)
        assert ei.value.needed == len(want)
        assert jpeg.encode_device(d_px, o) == want  # (the context's own pinned buffer, then Python bytes)


def test_encode_device_into_pinned_storage_every_alignment():
    _pinned_storage_cases()


def test_direct_store_switch_gives_the_same_files():
    """PIXO_HIP_DEBUG=direct_stores (the stuffing kernel writes the caller's pinned memory itself) in a fresh process:
    the alignment cases above, and the same bytes as the default path."""
    import subprocess, sys, hashlib
    code = ("import sys, hashlib, torch; sys.path.insert(0, 'tests'); import synth; from pixo_amd import jpeg\n"
            "px = synth.noise(1024, 768, 5); d = torch.from_numpy(px).to('cuda:0'); torch.cuda.synchronize()\n"
            "o = jpeg.JpegOptions.builder(1024, 768).quality(80).subsampling(jpeg.Subsampling.S420).build()\n"
            "pin = torch.zeros(4 << 20, dtype=torch.uint8).pin_memory(); n = jpeg.encode_device_into(pin, d, o)\n"
            "print(hashlib.sha256(pin[:n].numpy().tobytes()).hexdigest(), hashlib.sha256(jpeg.encode_device(d, o)).hexdigest())")
    import os
    outs = []
    for direct in ("", "direct_stores"):
        env = dict(os.environ, PIXO_HIP_DEBUG=direct)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if len(ln.split()) == 2 and len(ln.split()[0]) == 64][-1].split())
    px = synth.noise(1024, 768, 5)
    o = jpeg.JpegOptions.builder(1024, 768).quality(80).subsampling(jpeg.Subsampling.S420).build()
    want = hashlib.sha256(jpeg.encode(px, o)).hexdigest()
    assert outs[0] == [want, want] and outs[1] == [want, want]
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, 'tests'); import test_gpu_parity as t; t._pinned_storage_cases(); print('cases ok')"],
                       capture_output=True, text=True, env=dict(os.environ, PIXO_HIP_DEBUG="direct_stores"), timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "cases ok" in r.stdout, r.stderr[-2000:]


def test_stuffing_grid_guesses_that_fall_short_are_completed():
    """The stuffing kernel's grid is sized for 64 bytes per block before the scan's length is known: noise at q = 100
    (150 bytes per block) needs more than twice as many tiles — the missing ones are launched afterwards, same bytes;
    smooth images in between use a fraction of the grid."""
    w, h = 1024, 768
    smooth, noisy = synth.gradient_rgb(w, h), synth.noise(w, h, 19)
    for px, q in ((smooth, 60), (noisy, 100), (smooth, 60), (noisy, 35), (noisy, 100), (smooth, 95)):
        for ss in (1, 0):
            o = jpeg.JpegOptions.builder(w, h).quality(q).subsampling(jpeg.Subsampling(ss)).build()
            assert jpeg.encode(px, o) == O.encode(px, O.make_options(w, h, 2, q, ss)), (q, ss)


def test_scans_coded_in_pieces_give_the_same_files():
    """Large scans are coded in pieces (runs of groups, one launch pair each) whose bytes leave for the host while the next
    piece is coded; the pieces hand each other bit and byte positions on the device.  With PIXO_HIP_DEBUG=piece_groups=1 and 3
    every reference-made golden above two or six groups goes through that path in up to 16 pieces — every alignment of
    the seams — and PIXO_HIP_DEBUG=one_piece switches it off; a 4096x4096 4:4:4 image takes the path by its own size."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env in ({"PIXO_HIP_DEBUG": "piece_groups=1"}, {"PIXO_HIP_DEBUG": "piece_groups=3"}, {"PIXO_HIP_DEBUG": "one_piece"},
                {"PIXO_HIP_DEBUG": "piece_medium=2"}, {"PIXO_HIP_DEBUG": "piece_medium=3,piece_schedule=1:2:5"}):
        r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                            "-k", "goldens or device_entropy_stage or encode_device_into_pinned_and_pageable or band"],
                           capture_output=True, text=True, env=dict(os.environ, **env), timeout=900, cwd=root)
        assert r.returncode == 0, (env, r.stdout[-3000:])
    # a piece whose stream outgrows the guess its stuffing grid was sized for (64 bytes per block; noise at q = 100 has
    # 150) sends the call back to the one-piece path: same bytes
    code = ("import sys; sys.path.insert(0, 'tests'); import synth, oracle_lib as O; from pixo_amd import jpeg\n"
            "px = synth.noise(256, 256, 3)\n"
            "for ss in (0, 1):\n"
            "    o = jpeg.JpegOptions.builder(256, 256).quality(100).subsampling(jpeg.Subsampling(ss)).build()\n"
            "    assert jpeg.encode(px, o) == O.encode(px, O.make_options(256, 256, 2, 100, ss))\n"
            "print('redo ok')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PIXO_HIP_DEBUG="piece_groups=1"),
                       timeout=300, cwd=root)
    assert r.returncode == 0 and "redo ok" in r.stdout, r.stderr[-2000:]
    # a medium scan (a 4096x4096 4:2:0 image: 2048 groups) is cut into growing pieces once the context has seen that its
    # files are large: the first call in one piece, the following ones in two — same bytes; a smooth image in between
    # switches back
    import torch
    w = h = 4096
    px = synth.noise(w, h, 42)
    o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420).build()
    want = O.encode(px, O.make_options(w, h, 2, 80, 1))
    d_px = torch.from_numpy(px).to("cuda:0")
    d_smooth = torch.from_numpy(synth.gradient_rgb(w, h)).to("cuda:0")
    smooth_want = O.encode(synth.gradient_rgb(w, h), O.make_options(w, h, 2, 80, 1))
    pinned = torch.full((w * h * 3,), 0x44, dtype=torch.uint8).pin_memory()
    for d, ref in ((d_px, want), (d_px, want), (d_px, want), (d_smooth, smooth_want), (d_px, want), (d_px, want)):
        n = jpeg.encode_device_into(pinned, d, o)
        assert n == len(ref) and pinned[:n].numpy().tobytes() == ref
        assert jpeg.encode_device(d, o) == ref  # (into a block of its own)
    # by its own size: 4096 groups of 192 blocks = two pieces (and the 16384x16384 file of test_config4... = sixteen)
    w, h = 4096, 4096
    px = synth.noise(w, h, 77)
    o = jpeg.JpegOptions.builder(w, h).quality(75).subsampling(jpeg.Subsampling.S444).build()
    want = O.encode(px, O.make_options(w, h, 2, 75, 0))
    assert jpeg.encode(px, o) == want
    d_px = torch.from_numpy(px).to("cuda:0")
    pinned = torch.full((w * h * 3,), 0x33, dtype=torch.uint8).pin_memory()
    n = jpeg.encode_device_into(pinned, d_px, o)
    assert n == len(want) and pinned[:n].numpy().tobytes() == want and bool((pinned[n:] == 0x33).all())


def test_every_rgb_colour_once():
    """A 4096x4096 image that contains each of the 16,777,216 RGB triples exactly once (two different
    arrangements, so that every colour meets different neighbours in the 4:2:0 box sums): coefficient
    tuples bit-exact against the oracle for 4:4:4 and 4:2:0."""
    w = h = 4096
    i = np.arange(w * h, dtype=np.uint32)
    for arrangement in range(2):
        k = i if arrangement == 0 else (i * np.uint32(2654435761)) & np.uint32(0xFFFFFF)  # odd multiplier: a permutation
        px = np.stack([(k & 255), (k >> 8) & 255, (k >> 16) & 255], axis=1).astype(np.uint8).reshape(-1)
        assert len(np.unique(k)) == w * h
        for ss in (0, 1):
            _check_coeffs(px, w, h, 2, ss, 92)


def test_restart_files_from_the_gpu_decode_like_the_plain_files():
    """Independent decoder (Pillow / libjpeg) on the DEVICE coder's output: the file with restart markers and the file
    without them hold the same coefficients, so they must decode to the same pixels (restart intervals cannot be
    produced by the reference's wasm entry; see tests/test_independent_decoders.py)."""
    import io
    from PIL import Image
    for (w, h, ct, ss) in [(333, 211, 2, 1), (200, 120, 2, 0), (129, 65, 0, 0)]:
        px = synth.noise_gray(w, h, 4) if ct == 0 else synth.gradient_rgb(w, h) ^ (synth.noise(w, h, 4) >> 3)
        plain = np.asarray(Image.open(io.BytesIO(jpeg.encode(px, _opts(w, h, ct, ss, 85)))))
        for interval in (1, 3, 8, 50):
            blob = jpeg.encode(px, _opts(w, h, ct, ss, 85, restart_interval=interval))
            assert blob.count(b"\xff\xdd") >= 1
            assert np.array_equal(np.asarray(Image.open(io.BytesIO(blob))), plain), (w, h, ct, ss, interval)


@pytest.mark.parametrize("case", [(64, 64, 2, 80), (333, 211, 2, 35), (1000, 37, 2, 100), (129, 65, 0, 90), (8, 8, 2, 1), (1, 1, 0, 50)])
def test_integer_mode_kernel_equals_the_oracle(case):
    """the labelled integer secondary mode (pixo_hip_jpeg_coeffs_integer) against the restatement of the
    reference's fixed-point family (oracle/pixo_int_oracle.c; pinned on the reference's unit-test values only — the
    family is dead code upstream).  Noise, flat images (the constant-block shortcut of dct_2d_fast) and gradients."""
    from pixo_amd import error
    w, h, ct, q = case
    for px in ((synth.noise_gray(w, h, 3) if ct == 0 else synth.noise(w, h, 3)),
               np.full(w * h * (1 if ct == 0 else 3), 200, np.uint8),
               (synth.gradient_rgb(w, h)[: w * h] if ct == 0 else synth.gradient_rgb(w, h))):
        gy, gcb, gcr = jpeg.coefficients_integer(px, _opts(w, h, ct, 0, q))
        oy, ocb, ocr = O.coeffs_integer(px, w, h, ct, q)
        assert np.array_equal(gy, oy) and np.array_equal(gcb, ocb) and np.array_equal(gcr, ocr)
    if ct == 2:
        with pytest.raises(error.Error, match="4:4:4 or gray only"):
            jpeg.coefficients_integer(px, _opts(w, h, 2, 1, q))


@pytest.mark.gpu
def test_large_batches_go_in_sub_batches_over_two_contexts_and_give_the_same_files():
    """A batch of 64 MB of pixels and more is cut into sub-batches that alternate between two contexts, so that one
    sub-batch's files cross PCIe while the next one's kernels run: same files at the same places as one image at a
    time, for batch sizes that do and do not divide evenly, with a smooth image among the noise and a short arena."""
    import torch
    from pixo_amd import error
    w, h = 1280, 720
    o = _opts(w, h, 2, 1, 80)
    base = [synth.noise(w, h, 900 + i) for i in range(4)] + [synth.gradient_rgb(w, h)]
    want_of = [O.encode(im, O.make_options(w, h, 2, 80, 1)) for im in base]
    for n in (25, 33, 64):  # 3, 4 and 8 sub-batches
        order = [(7 * i + 3) % len(base) for i in range(n)]
        d_px = torch.from_numpy(np.concatenate([base[k] for k in order])).to("cuda:0")
        torch.cuda.synchronize()
        want = [want_of[k] for k in order]
        total = sum(len(f) for f in want)
        arena = torch.full((total + 32,), 0x5A, dtype=torch.uint8).pin_memory()
        offs, lens = jpeg.encode_batch_device_into(arena, d_px, o, n)
        assert lens == [len(f) for f in want] and offs == [sum(lens[:i]) for i in range(n)]
        raw = arena.numpy()
        for i in range(n):
            assert raw[offs[i]: offs[i] + lens[i]].tobytes() == want[i], (n, i)
        assert bool((raw[total:] == 0x5A).all())
        with pytest.raises(error.Error, match="need %d bytes" % total):
            jpeg.encode_batch_device_into(torch.zeros(total - 1, dtype=torch.uint8).pin_memory(), d_px, o, n)
        files = jpeg.encode_batch_device(d_px, o, n)  # (the malloc'ing form: one pass, unchanged)
        assert files == want


@pytest.mark.gpu
def test_config_3_exactly_all_64_files_of_64_x_1080p_against_the_oracle():
    """BASELINE configs[2] at its exact shape: 64 x 1920x1080 RGB8 noise, seeds 42..105, q=80 4:2:0, device resident, into one
    pinned arena in one call — ALL 64 files equal the oracle's, file 0 equals the reference-made golden of SURVEY §8c, and the
    same batch with the files left in HBM (device arena, what sharded.encode_batch gathers over RCCL) gives the same bytes."""
    import hashlib
    import torch
    w, h, n = 1920, 1080, 64
    o = _opts(w, h, 2, 1, 80)
    oo = O.make_options(w, h, 2, 80, 1)
    imgs = [synth.noise(w, h, 42 + i) for i in range(n)]
    d_px = torch.from_numpy(np.concatenate(imgs)).to("cuda:0")
    torch.cuda.synchronize()
    arena = torch.empty(n * w * h, dtype=torch.uint8).pin_memory()
    offs, lens = jpeg.encode_batch_device_into(arena, d_px, o, n)
    raw = arena.numpy()
    assert hashlib.sha256(raw[: lens[0]].tobytes()).hexdigest() == "d1811ba1761f6b2a76d7f2c3d43418784f38909e0b20af631d5ead73e7d9436a"
    assert lens[0] == 1388296 and offs == [sum(lens[:i]) for i in range(n)]
    for i in range(n):
        assert raw[offs[i]: offs[i] + lens[i]].tobytes() == O.encode(imgs[i], oo), i
    total = offs[-1] + lens[-1]
    d_arena = torch.full((total + 16,), 0x5A, dtype=torch.uint8, device="cuda:0")
    offs2, lens2 = jpeg.encode_batch_device_into(d_arena, d_px, o, n)
    assert (offs2, lens2) == (offs, lens)
    back = d_arena.cpu().numpy()
    assert np.array_equal(back[:total], raw[:total]) and bool((back[total:] == 0x5A).all())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(333, 77, 2, 0, 9, {}), (64, 64, 0, 0, 5, {}), (100, 36, 2, 1, 3, {"optimize_huffman": True}),
                                   (200, 120, 2, 1, 4, {"restart_interval": 3}), (16, 16, 2, 1, 40, {"progressive": True})])
def test_batch_into_a_device_arena_files_complete_in_hbm(shape):
    """`arena` in device memory: headers, scans and EOI of every file at their final places in HBM, for the one-pass batch and
    for option sets that are coded image by image; one byte short is refused with the size needed."""
    import torch
    from pixo_amd import error
    w, h, ct, ss, n, kw = shape
    imgs = [(synth.noise(w, h, 700 + i) if ct == 2 else synth.noise_gray(w, h, 700 + i)) for i in range(n)]
    okw = {("restart" if k == "restart_interval" else k): v for k, v in kw.items()}
    want = [O.encode(im, O.make_options(w, h, ct, 80, ss, **okw)) for im in imgs]
    total = sum(len(f) for f in want)
    d_px = torch.from_numpy(np.concatenate(imgs)).to("cuda:0")
    d_arena = torch.full((total + 5,), 0x5A, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    o = _opts(w, h, ct, ss, 80, **kw)
    offs, lens = jpeg.encode_batch_device_into(d_arena, d_px, o, n)
    raw = d_arena.cpu().numpy()
    assert lens == [len(f) for f in want]
    for i in range(n):
        assert raw[offs[i]: offs[i] + lens[i]].tobytes() == want[i], i
    assert bool((raw[total:] == 0x5A).all())
    with pytest.raises(error.BufferTooSmall) as ei:
        jpeg.encode_batch_device_into(torch.empty(total - 1, dtype=torch.uint8, device="cuda:0"), d_px, o, n)
    assert "need %d bytes" % total in str(ei.value)


@pytest.mark.parametrize("w,h", [(65535, 1), (1, 65535), (65535, 9), (9, 65535), (65535, 17), (17, 65535)])
@pytest.mark.parametrize("preset", [0, 1, 2])
def test_maximum_dimension_strips_whole_files(w, h, preset):
    """The format's largest side (jpeg/mod.rs:338-345, MAX_DIMENSION) as a strip in either direction: 8,192 MCU columns / rows of
    which the last is partial, for the three wasm presets, RGB 4:2:0 / 4:4:4 and gray."""
    for ct, s420, seed in ((2, True, 5), (2, False, 6), (0, False, 7)):
        px = synth.noise_gray(w, h, seed) if ct == 0 else synth.noise(w, h, seed)
        want = bytes(O.encode_flat(px, w, h, ct, 77, preset, s420))
        got = bytes(jpeg.encode_jpeg(px, w, h, ct, 77, preset, s420))
        assert got == want, (w, h, preset, ct, s420)


def test_maximum_width_many_rows_and_maximum_height_many_columns():
    for (w, h) in ((65535, 200), (200, 65535)):
        px = synth.gradient_rgb(w, h)
        px = (px.astype(np.int32) + (synth.noise(w, h, 3).astype(np.int32) & 7)).clip(0, 255).astype(np.uint8)
        for s420 in (True, False):
            want = bytes(O.encode_flat(px, w, h, 2, 85, 0, s420))
            assert bytes(jpeg.encode_jpeg(px, w, h, 2, 85, 0, s420)) == want, (w, h, s420)


def test_plain_host_switch_keeps_no_blocks_and_the_bytes():
    """PIXO_HIP_DEBUG=plain_host: pixo_hip_free gives large blocks straight back (nothing cached), no madvise; the file is the same."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import hashlib, synth
from pixo_amd import jpeg
px = synth.noise(8192, 4096, 9)
o = jpeg.JpegOptions.builder(8192, 4096).quality(90).subsampling(jpeg.Subsampling(0)).build()
a = jpeg.encode(px, o); b = jpeg.encode(px, o)
assert a == b and len(a) > (24 << 20)
print("SHA", hashlib.sha256(a).hexdigest())
''' % (root, root)
    outs = []
    for dbg in ("", "plain_host"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, PIXO_HIP_DEBUG=dbg))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("SHA")][0])
    assert outs[0] == outs[1]


@pytest.mark.parametrize("shape", [(64, 64), (512, 384), (1024, 1024)])
def test_small_file_into_pinned_storage_that_is_too_small_exact_and_roomy(shape):
    """Small files are stored by the stuffing kernel straight into the caller's pinned buffer (round 4): a buffer that is too small by
    one byte, by half, or has room for 16 bytes only must give BufferTooSmall with the file's length and leave everything BEHIND its
    capacity alone; the exact size and a roomy buffer must give the file."""
    import torch
    from pixo_amd import error
    w, h = shape
    px = synth.noise(w, h, 31)
    o = _opts(w, h, 2, 1, 88)
    want = bytes(O.encode(px, O.make_options(w, h, 2, 88, 1)))
    d = torch.from_numpy(px).to("cuda:0")
    torch.cuda.synchronize()
    for cap in (16, len(want) // 2, len(want) - 1, len(want), len(want) + 4096):
        buf = torch.full((cap + 64,), 0xA5, dtype=torch.uint8).pin_memory()
        view = buf[:cap]
        if cap < len(want):
            with pytest.raises(error.BufferTooSmall) as ei:
                jpeg.encode_device_into(view, d, o)
            assert ei.value.needed == len(want)
        else:
            n = jpeg.encode_device_into(view, d, o)
            assert n == len(want) and view[:n].numpy().tobytes() == want
        assert bool((buf[cap:] == 0xA5).all()), "bytes behind the buffer's capacity were written (capacity %d)" % cap


def test_batch_in_sub_batches_gives_the_same_arena():
    """Round 5: pixo_hip_jpeg_encode_batch_device_into cuts a large batch into sub-batches (copies under the next one's kernels) —
    how many, it chooses from the content's bytes per block.  Whatever the count (debug switch batch_parts), the arena, the
    offsets and the lengths are the same, and the files are the oracle's."""
    import hashlib
    import torch
    w, h, n = 1920, 1080, 40  # (237 MB of pixels: above the 64 MB from which the library considers sub-batches)
    px = synth.photo(w, h, 5)
    d = torch.from_numpy(np.ascontiguousarray(px)).cuda().repeat(n).contiguous()
    d[w * h * 3 * 7: w * h * 3 * 8] ^= 0x10  # (one image of the batch differs)
    o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420).build()
    arena = torch.empty(n * w * h, dtype=torch.uint8).pin_memory()
    seen = {}
    try:
        for parts in (None, 1, 3, 8, None):
            jpeg.debug_configure("batch_parts=%d" % parts if parts else None)
            arena.zero_()
            offs, lens = jpeg.encode_batch_device_into(arena, d, o, n)
            end = offs[-1] + lens[-1]
            seen[parts if parts else 0] = (list(offs), list(lens), hashlib.sha256(arena[:end].numpy().tobytes()).hexdigest())
    finally:
        jpeg.debug_configure(None)
    assert len({repr(v) for v in seen.values()}) == 1, seen.keys()
    offs, lens, _ = seen[1]
    want = O.encode(px, O.make_options(w, h, 2, 80, 1))
    assert arena[offs[0]: offs[0] + lens[0]].numpy().tobytes() == want
    assert arena[offs[n - 1]: offs[n - 1] + lens[n - 1]].numpy().tobytes() == want
    other = bytes(d[w * h * 3 * 7: w * h * 3 * 8].cpu().numpy())
    assert arena[offs[7]: offs[7] + lens[7]].numpy().tobytes() == O.encode(np.frombuffer(other, np.uint8), O.make_options(w, h, 2, 80, 1))


def test_optimised_tables_with_restart_intervals_behind_a_standard_tables_call():
    """Found by tools/stress_parity.py (seed 955, a GPU memory fault): the pinned block that receives the segments' ends also receives the
    536 symbol counters of an optimised-tables pass, and growing it for the counters moved it under the address a segmented job had already
    handed to the stuffing kernel — only when an earlier call of the thread had left the block small.  The sequence, several sizes."""
    for (w, h, ss, restart) in ((95, 260, 0, 35), (95, 260, 0, 36), (640, 480, 1, 17), (333, 222, 0, 100)):
        n = w * h * 3
        px = synth.lcg_bytes(n, 1234).copy()
        px[px < 128] = 0
        px[px >= 128] = 255
        jpeg.trim()  # (the thread's buffers start over: the block is small again)
        for opt, r in ((False, None), (True, restart), (False, restart), (True, restart)):
            o = _opts(w, h, 2, ss, 77, **({"restart_interval": r} if r else {}), **({"optimize_huffman": True} if opt else {}))
            kw = dict(optimize_huffman=opt)
            if r:
                kw["restart"] = r
            assert jpeg.encode(px, o) == O.encode(px, O.make_options(w, h, 2, 77, ss, **kw)), (w, h, ss, opt, r)
