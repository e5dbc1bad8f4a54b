"""GPU parity of the fused pixel -> bit stream kernel (pixo_amd/csrc/jpeg_pixels_code.hip; round 5): whole baseline files
through the C ABI must equal the oracle's bytes — and the bytes of the two-kernel form (coefficient kernel + scan_code,
debug switch `two_kernel_scan`) — on tile-edge sizes, both subsamplings, long blocks (noise at q = 100), smooth content
(groups of a few bits), saturated colours, host pixels and device pixels."""
import numpy as np
import pytest

import oracle_lib as O
import synth
from pixo_amd import ColorType, jpeg

pytestmark = pytest.mark.gpu


def _opts(w, h, ss, q):
    return jpeg.JpegOptions.builder(w, h).color_type(ColorType(2)).quality(q).subsampling(jpeg.Subsampling(ss)).build()


def _both(px, w, h, ss, q):
    o = _opts(w, h, ss, q)
    want = O.encode(px, O.make_options(w, h, 2, q, ss))
    got = jpeg.encode(px, o)
    assert got == want, "fused kernel: %dx%d ss=%d q=%d differs from the oracle (%d vs %d bytes)" % (w, h, ss, q, len(got), len(want))
    jpeg.debug_configure("two_kernel_scan")
    try:
        two = jpeg.encode(px, o)
    finally:
        jpeg.debug_configure(None)
    assert two == want


SIZES = [(4, 4), (5, 3), (16, 16), (17, 1), (31, 33), (511, 16), (512, 16), (513, 17), (528, 32), (1023, 48), (1024, 64), (1025, 8),
         (1536, 40), (2048, 16), (1918, 70), (1921, 40), (4096, 32), (4100, 24), (777, 555)]


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("ss", [1, 0])
def test_fused_kernel_edge_sizes(w, h, ss):
    _both(synth.noise(w, h, 7 + w + h), w, h, ss, 80)


@pytest.mark.parametrize("q", [1, 10, 50, 80, 95, 100])
@pytest.mark.parametrize("ss", [1, 0])
def test_fused_kernel_quality_and_content(q, ss):
    w, h = 1100, 200
    for px in (synth.noise(w, h, q), synth.gradient_rgb(w, h), synth.flat_blocks(w, h), synth.checkerboard(w, h, 5), synth.extremes(w, h, 3)):
        _both(px, w, h, ss, q)


def test_fused_kernel_flat_image_groups_of_a_few_bits():
    """a constant image: every block is 2 + 2 or 4 bits — groups of 768 bits, many sharing stream words"""
    for (w, h) in [(2048, 256), (600, 40)]:
        for rgb in ([0, 0, 255], [255, 0, 0], [17, 200, 3], [128, 128, 128]):
            px = np.tile(np.array(rgb, np.uint8), w * h)
            _both(px, w, h, 1, 80)
            _both(px, w, h, 0, 100)


def test_fused_kernel_4096_square_device_pixels():
    """configs[1]'s image, device resident, into a pinned buffer — the path bench.py's whole_file times"""
    import torch
    w = h = 4096
    for gen, q in ((synth.noise, 80), (synth.gradient_rgb, 80)):
        px = gen(w, h, 42) if gen is synth.noise else gen(w, h)
        o = _opts(w, h, 1, q)
        want = O.encode(px, O.make_options(w, h, 2, q, 1))
        d = torch.from_numpy(np.ascontiguousarray(px)).cuda()
        buf = torch.empty(len(want) + 4096, dtype=torch.uint8).pin_memory()
        for _ in range(3):  # (the context's state buffers are reused: clean after every file)
            n = jpeg.encode_device_into(buf, d, o)
            assert bytes(buf[:n].numpy().tobytes()) == want
        assert jpeg.encode_device(d, o) == want


def test_fused_kernel_tall_narrow_and_many_rows():
    for (w, h) in [(16, 4096), (40, 3000), (520, 1500)]:
        _both(synth.noise(w, h, 5), w, h, 1, 85)
        _both(synth.noise(w, h, 6), w, h, 0, 60)


def test_pinned_destination_one_byte_short_is_never_written_beyond_its_capacity():
    """small files are stored into a pinned destination by the stuffing kernel itself.  A destination that is one
    byte short (or holds only the headers + 2 bytes) must get BufferTooSmall with the size needed, and the guard bytes behind
    its capacity must stay untouched."""
    import torch
    from pixo_amd import error
    w, h = 256, 160
    px = synth.noise(w, h, 9)
    o = _opts(w, h, 1, 80)
    want = O.encode(px, O.make_options(w, h, 2, 80, 1))
    d = torch.from_numpy(np.ascontiguousarray(px)).cuda()
    for cap in (len(want) - 1, 700, 625):
        pinned = torch.full((len(want) + 64,), 0xA5, dtype=torch.uint8).pin_memory()
        with pytest.raises(error.BufferTooSmall) as e:
            jpeg.encode_device_into(pinned[:cap], d, o)
        assert e.value.needed == len(want)
        assert bool((pinned[cap:] == 0xA5).all()), "bytes behind the capacity were written"
    pinned = torch.full((len(want) + 64,), 0xA5, dtype=torch.uint8).pin_memory()
    n = jpeg.encode_device_into(pinned[: len(want)], d, o)
    assert n == len(want) and pinned[:n].numpy().tobytes() == want and bool((pinned[n:] == 0xA5).all())


# ---- round 6: segments — the images of a batch and restart intervals of whole MCU rows go through the fused kernel --------------
def _form(d, o, batch=1):
    """1: the fused kernel serves this job, 0: coefficient kernel + scan_code + stuffing kernel"""
    import torch
    f = jpeg.debug_scan_device_async(d, o, stream=torch.cuda.current_stream().cuda_stream, batch=batch)
    torch.cuda.synchronize()
    return f


def _batch_images(w, h, n, seed):
    gens = [lambda i: synth.noise(w, h, seed + i), lambda i: synth.gradient_rgb(w, h), lambda i: synth.photo(w, h, seed + i),
            lambda i: synth.flat_blocks(w, h), lambda i: np.tile(np.array([255, 255, 255], np.uint8), w * h),
            lambda i: synth.extremes(w, h, seed + i)]
    return [np.ascontiguousarray(gens[i % len(gens)](i)).reshape(-1) for i in range(n)]


@pytest.mark.parametrize("w,h,n,ss,q", [(640, 48, 5, 1, 80), (1920, 1080, 3, 1, 80), (96, 64, 70, 1, 75), (16, 16, 300, 1, 90), (1100, 200, 4, 0, 80),
                                        (520, 24, 9, 0, 50), (2048, 32, 2, 1, 100), (36, 20, 33, 0, 100), (4096, 2048, 2, 1, 80), (16, 16, 600, 1, 85)])
def test_fused_kernel_batches_every_file_equals_the_oracle(w, h, n, ss, q):
    """a batch = one launch of the fused kernel with every image a segment (chains of their own, a look-back over the segments'
    byte counts for where each file begins): host arena, device arena, malloc'd files — all against the oracle, and against the
    two-kernel form"""
    import torch
    imgs = _batch_images(w, h, n, 11)
    d = torch.from_numpy(np.concatenate(imgs)).cuda()
    o = _opts(w, h, ss, q)
    want = [O.encode(px, O.make_options(w, h, 2, q, ss)) for px in imgs]
    total = sum(len(f) for f in want)
    # (switch fused_batch: the fused kernel whatever the width; by default batches of images whose 512-pixel tiles are at least three
    # quarters full take it — scan_job.cpp pixels_code_usable — and the others coefficient kernel + scan_code + stuffing kernel)
    jpeg.debug_configure("fused_batch")
    try:
        assert _form(d, o, n) == 1, "the batch did not take the fused kernel"
        arena = torch.empty(total + 64, dtype=torch.uint8).pin_memory()
        for _ in range(2):  # (the context's alternating state blocks: clean after every launch)
            offs, lens = jpeg.encode_batch_device_into(arena, d, o, n)
            a = arena.numpy()
            for i in range(n):
                assert a[offs[i]: offs[i] + lens[i]].tobytes() == want[i], "file %d of the batch differs from the oracle" % i
        darena = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
        offs, lens = jpeg.encode_batch_device_into(darena, d, o, n)
        a = darena.cpu().numpy()
        for i in range(n):
            assert a[offs[i]: offs[i] + lens[i]].tobytes() == want[i]
        files = jpeg.encode_batch_device(d, o, n)
        assert [bytes(f) for f in files] == want
    finally:
        jpeg.debug_configure(None)
    unit, per_tile = (16, 32) if ss == 1 else (8, 64)
    units_x = (w + unit - 1) // unit
    tiles_x = (units_x + per_tile - 1) // per_tile
    by_default = 1 if units_x * 4 >= tiles_x * per_tile * 3 else 0
    if by_default or w * h >= 96 * 64:  # (the measuring entry wants a single-pass job: images of 96 blocks and more for the tuple kernels)
        assert _form(d, o, n) == by_default
    assert [bytes(f) for f in jpeg.encode_batch_device(d, o, n)] == want
    jpeg.debug_configure("two_kernel_scan")
    try:
        assert [bytes(f) for f in jpeg.encode_batch_device(d, o, n)] == want
    finally:
        jpeg.debug_configure(None)
    assert jpeg.lookback_fallbacks() == 0


@pytest.mark.parametrize("w,h,ss,rows", [(640, 200, 1, 1), (640, 200, 1, 3), (1100, 333, 0, 2), (4096, 512, 1, 1), (513, 100, 1, 7), (200, 4000, 0, 5),
                                         (64, 64, 1, 1), (2048, 2048, 1, 9), (64, 8192, 0, 1)])  # (the last: 1024 segments — block sums beyond one 4 KiB copy)
def test_fused_kernel_restart_intervals_of_whole_mcu_rows(w, h, ss, rows):
    """restart_interval = rows x (MCUs per row): every interval a segment of the fused kernel, RSTn written by the segment's last
    group (jpeg/mod.rs:1423-1445); an interval that is NOT whole rows keeps the two-kernel form.  (The restart branch is pinned by
    the oracle only: the reference's wasm entry cannot pass restart_interval.)"""
    import torch
    unit = 16 if ss == 1 else 8
    units_x = (w + unit - 1) // unit
    px = synth.noise(w, h, 5 + rows) if rows % 2 else synth.photo(w, h, 5 + rows)
    d = torch.from_numpy(np.ascontiguousarray(px)).cuda()
    for interval, fused in ((rows * units_x, 1), (rows * units_x + 1, 0)):
        if interval > 65535:
            continue
        o = jpeg.JpegOptions.builder(w, h).color_type(ColorType(2)).quality(80).subsampling(jpeg.Subsampling(ss)).restart_interval(interval).build()
        units = units_x * ((h + unit - 1) // unit)
        if interval < units and (fused or interval * (6 if ss else 3) >= 96):
            assert _form(d, o) == fused
        want = O.encode(px, O.make_options(w, h, 2, 80, ss, restart=interval))
        assert jpeg.encode_device(d, o) == want, "restart interval %d: file differs from the oracle" % interval
        assert jpeg.encode(px, o) == want
    assert jpeg.lookback_fallbacks() == 0


def test_fused_kernel_batch_with_long_groups_and_small_output():
    """noise at q = 100: groups of several rounds (the blocks parked, walked twice) inside a batch; and an arena that is too small
    reports the size needed without a byte behind its capacity being written"""
    import torch
    from pixo_amd import error
    w, h, n = 1030, 40, 4
    imgs = [synth.noise(w, h, 70 + i) for i in range(n)]
    d = torch.from_numpy(np.concatenate(imgs)).cuda()
    jpeg.debug_configure("fused_batch")
    for ss in (1, 0):
        o = _opts(w, h, ss, 100)
        want = [O.encode(px, O.make_options(w, h, 2, 100, ss)) for px in imgs]
        total = sum(len(f) for f in want)
        arena = torch.full((total + 64,), 0xA5, dtype=torch.uint8).pin_memory()
        offs, lens = jpeg.encode_batch_device_into(arena[:total], d, o, n)
        a = arena.numpy()
        for i in range(n):
            assert a[offs[i]: offs[i] + lens[i]].tobytes() == want[i]
        assert bool((arena[total:] == 0xA5).all())
        small = torch.full((total,), 0xA5, dtype=torch.uint8).pin_memory()
        with pytest.raises(error.BufferTooSmall) as e:
            jpeg.encode_batch_device_into(small[: total - 1], d, o, n)
        assert e.value.needed == total and int(small[total - 1]) == 0xA5
    jpeg.debug_configure(None)


@pytest.mark.parametrize("w,h,threads", [(4096, 4096, 4), (1920, 1080, 6), (640, 480, 8)])
def test_calling_threads_start_single_pass_kernels_together_without_starving_each_other(w, h, threads):
    """Several threads, each with a context and a stream of its own, start together (pixo's rayon callers: src/jpeg/mod.rs:88 from a
    par_iter).  Two single-pass launches that split an empty device's workgroup slots between them wait for each other until the
    bounded waits give up (0.75 s, then the multi-pass kernels): the dispatch gate (pixo_amd/csrc/dispatch_gate.hpp) keeps ONE
    such launch at a time in its dispatch phase unless they fit the device together.  Every file = the oracle's, no fallback."""
    import threading
    import torch
    px = synth.photo(w, h, 17)
    d = torch.from_numpy(np.ascontiguousarray(px)).cuda()
    o = jpeg.JpegOptions.builder(w, h).color_type(ColorType(2)).quality(80).subsampling(jpeg.Subsampling(1)).build()
    want = O.encode(px, O.make_options(w, h, 2, 80, 1))
    fb0 = jpeg.lookback_fallbacks()
    for two_kernel in (False, True):
        jpeg.debug_configure("two_kernel_scan" if two_kernel else None)
        try:
            for rep in range(3):  # (fresh threads every time: they adopt parked contexts and start at the same moment)
                gate = threading.Barrier(threads)
                bad = []

                def work():
                    gate.wait()
                    for _ in range(4):
                        if jpeg.encode_device(d, o) != want:
                            bad.append(1)
                ts = [threading.Thread(target=work) for _ in range(threads)]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
                assert not bad
        finally:
            jpeg.debug_configure(None)
    assert jpeg.lookback_fallbacks() == fb0
    waits, timeouts = jpeg.dispatch_gate_stats()  # (a wait that ran into its 5 ms bound is allowed — the launch then goes ahead and the
    assert waits >= 0 and timeouts >= 0           #  kernels' bounded waits remain — what must not happen is a fallback or a wrong file)


@pytest.mark.parametrize("w,h,ss,q,rows", [(640, 480, 1, 80, 0), (640, 480, 0, 75, 0), (1030, 37, 1, 90, 0), (20, 20, 0, 50, 0), (4096, 512, 1, 80, 0),
                                           (2048, 2048, 0, 85, 0), (1100, 333, 0, 80, 2), (640, 200, 1, 80, 3), (4096, 4096, 1, 80, 0), (4094, 1000, 1, 80, 0)])
def test_optimised_tables_statistics_from_the_pixels(w, h, ss, q, rows):
    """optimize_huffman through the fused kernel: the statistics come from the pixels as well (launch_pixels_count — the reference runs
    its pixel pipeline twice for this preset too, src/jpeg/mod.rs:826-860 then :1408-1563), the DC symbols of the tiles' first blocks from
    the (last DC of the tile before, first DC of this tile) pairs; restart intervals of whole MCU rows reset the predictors.  Files =
    the oracle's, and = the two-kernel form's, for noise and photograph-like content."""
    import torch
    unit = 16 if ss == 1 else 8
    units_x = (w + unit - 1) // unit
    for px in (synth.noise(w, h, 31 + w), synth.photo(w, h, 32 + h)):
        d = torch.from_numpy(np.ascontiguousarray(px)).cuda()
        b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(2)).quality(q).subsampling(jpeg.Subsampling(ss)).optimize_huffman(True)
        if rows:
            b = b.restart_interval(rows * units_x)
        o = b.build()
        want = O.encode(px, O.make_options(w, h, 2, q, ss, optimize_huffman=True, restart=rows * units_x if rows else None))
        assert jpeg.encode_device(d, o) == want, "optimised tables, device pixels: file differs from the oracle"
        assert jpeg.encode(px, o) == want, "optimised tables, host pixels: file differs from the oracle"
        jpeg.debug_configure("two_kernel_scan")
        try:
            assert jpeg.encode_device(d, o) == want
        finally:
            jpeg.debug_configure(None)
    assert jpeg.lookback_fallbacks() == 0


@pytest.mark.parametrize("w,h,q", [(1536, 8, 80), (1537, 9, 80), (4, 1, 75), (5, 5, 90), (200, 120, 80), (1535, 64, 50), (3073, 33, 85), (4096, 4096, 80),
                                   (4094, 515, 80), (1920, 1080, 92), (6150, 24, 100)])
def test_gray_images_through_the_fused_kernel(w, h, q):
    """Gray8 (ColorType::Gray, src/jpeg/mod.rs:1448-1470): tiles of 1536 x 8 pixels — 192 consecutive blocks of one block row — through
    the fused kernel; rows of any alignment, partial last tiles, edge replication (extract_block :1565-1606), standard and optimised
    tables, restart intervals of whole block rows, a batch.  Files = the oracle's = the two-kernel form's."""
    import torch
    units_x = (w + 7) // 8
    for px in (synth.noise_gray(w, h, 7 + w), synth.photo(w, h, 8 + h).reshape(-1, 3)[:, 1].copy()):
        d = torch.from_numpy(np.ascontiguousarray(px)).cuda()
        for opt, rows in ((False, 0), (True, 0), (False, 2), (True, 1)):
            if rows and rows * units_x > 65535:
                continue
            b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(0)).quality(q).optimize_huffman(opt)
            if rows:
                b = b.restart_interval(rows * units_x)
            o = b.build()
            want = O.encode(px, O.make_options(w, h, 0, q, 0, optimize_huffman=opt, restart=rows * units_x if rows else None))
            if not opt:
                assert _form(d, o) == 1, "a gray image did not take the fused kernel"
            assert jpeg.encode_device(d, o) == want, "gray, device pixels (optimised %s, restart rows %d): file differs from the oracle" % (opt, rows)
            assert jpeg.encode(px, o) == want
            jpeg.debug_configure("two_kernel_scan")
            try:
                assert jpeg.encode_device(d, o) == want
            finally:
                jpeg.debug_configure(None)
    if w * h <= 1920 * 1080:  # a batch of gray images: every image a segment
        n = 5
        imgs = [synth.noise_gray(w, h, 50 + i) for i in range(n)]
        d = torch.from_numpy(np.concatenate(imgs)).cuda()
        o = jpeg.JpegOptions.builder(w, h).color_type(ColorType(0)).quality(q).build()
        want = [O.encode(px, O.make_options(w, h, 0, q, 0)) for px in imgs]
        jpeg.debug_configure("fused_batch")
        try:
            assert [bytes(f) for f in jpeg.encode_batch_device(d, o, n)] == want
        finally:
            jpeg.debug_configure(None)
        assert [bytes(f) for f in jpeg.encode_batch_device(d, o, n)] == want
    assert jpeg.lookback_fallbacks() == 0


@pytest.mark.parametrize("w,h,ct,ss", [(8, 212, 0, 0), (7, 230, 0, 0), (4, 4000, 0, 0), (8, 300, 2, 0), (5, 333, 2, 0), (16, 500, 2, 1), (1537, 100, 0, 0), (520, 90, 2, 0)])
def test_groups_of_fewer_than_seven_bits_hand_on_the_bits_in_front_of_them(w, h, ct, ss):
    """A tile with one block (narrow gray images, a row's last tile) under optimised tables whose codes are one bit long codes two or
    three bits per group: the seven bits a group hands to the group behind it are then not all its own (found by tools/stress_parity.py:
    gray 8 x 212, optimize_huffman, a slow ramp).  Ramps, flat images and noise; standard and optimised tables; restart rows."""
    import torch
    n = w * h * (3 if ct == 2 else 1)
    unit = 16 if ss == 1 else 8
    units_x = (w + unit - 1) // unit
    contents = [((np.arange(n, dtype=np.int64) // 3 // 70) % 256).astype(np.uint8), np.full(n, 200, np.uint8), synth.lcg_bytes(n, 9),
                ((np.arange(n, dtype=np.int64) // (3 * w)) % 256).astype(np.uint8)]
    for px in contents:
        d = torch.from_numpy(px).cuda()
        for opt in (True, False):
            for rows in (0, 3):
                for q in (37, 80):
                    b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss)).optimize_huffman(opt)
                    if rows:
                        b = b.restart_interval(rows * units_x)
                    o = b.build()
                    want = O.encode(px, O.make_options(w, h, ct, q, ss, optimize_huffman=opt, restart=rows * units_x if rows else None))
                    assert jpeg.encode_device(d, o) == want, (w, h, ct, ss, opt, rows, q)
    assert jpeg.lookback_fallbacks() == 0


def test_no_chain_through_all_groups_or_segments_of_a_launch():
    """Timing with a wide margin, not parity: three times this round a workgroup that WAITED before it let its own value out made
    every group (or segment) of a launch wait for the whole one before it — milliseconds where the work is tens of microseconds
    (profiles/r06_fused_batches_chain.txt, r06_long_groups_chain.txt).  Device time by events, best of three, against bounds 8-20 x
    above what the kernels take: (1) groups of several rounds (noise at q = 95: was 10 us per group), (2) the segments of a batch
    (was 6.5 us per segment on top), (3) restart rows."""
    import torch
    stream = torch.cuda.current_stream().cuda_stream

    def device_us(d, o, batch=1):
        best = 1e9
        for _ in range(3):
            for _ in range(3):
                jpeg.debug_scan_device_async(d, o, stream=stream, batch=batch)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                jpeg.debug_scan_device_async(d, o, stream=stream, batch=batch)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
        return best

    w = h = 2048  # 512 groups
    d = torch.from_numpy(synth.noise(w, h, 3)).cuda()
    o95 = jpeg.JpegOptions.builder(w, h).color_type(ColorType(2)).quality(95).subsampling(jpeg.Subsampling(1)).build()
    assert _form(d, o95) == 1
    t = device_us(d, o95)
    assert t < 1000, "groups of several rounds: %.0f us for 512 groups (a chain through the groups takes 5000)" % t
    g = torch.from_numpy(synth.noise_gray(w, h, 4)).cuda()
    og = jpeg.JpegOptions.builder(w, h).color_type(ColorType(0)).quality(90).build()
    t = device_us(g, og)
    assert t < 1000, "gray groups of several rounds: %.0f us" % t
    n, bw, bh = 128, 1024, 256  # 128 segments of 32 groups
    imgs = np.concatenate([synth.photo(bw, bh, 60 + i % 4) for i in range(n)])
    db = torch.from_numpy(imgs).cuda()
    ob = jpeg.JpegOptions.builder(bw, bh).color_type(ColorType(2)).quality(80).subsampling(jpeg.Subsampling(1)).build()
    assert _form(db, ob, n) == 1
    t = device_us(db, ob, n)
    assert t < 500, "a batch of 128 segments: %.0f us (a chain through the segments adds 800)" % t
    orst = jpeg.JpegOptions.builder(w, h).color_type(ColorType(2)).quality(80).subsampling(jpeg.Subsampling(1)).restart_interval(w // 16).build()
    t = device_us(d, orst)
    assert t < 500, "128 restart rows: %.0f us" % t
