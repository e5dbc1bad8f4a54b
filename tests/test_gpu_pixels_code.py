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
