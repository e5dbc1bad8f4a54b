"""The N>1 path on CPU (gloo, 127.0.0.1 rendezvous, world_size 2 and 3).

`pixo_amd.sharded.encode_banded` is the per-band-entropy form: every rank computes ITS
band's coefficients (here with the oracle standing in for the rank's GPU — the point is the sharding,
the exchanges and the splice, not the arithmetic), entropy-codes it with the product's host twins of
the band encoder, and the ranks exchange only boundary DCs (3 x i16), bit totals (u64) and — for
optimised tables — symbol counts; rank 0 splices the pieces.  `encode_gathered*` are the fallbacks that
ship coefficient bands to rank 0 (progressive scans, restart markers).  Every file must be
byte-identical to the oracle's single-image file."""
import os
import socket
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def _options(w, h, ct, ss, q, flags):
    from pixo_amd import ColorType, jpeg
    b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss))
    if flags.get("optimize_huffman"): b = b.optimize_huffman(True)
    if flags.get("progressive"): b = b.progressive(True)
    if flags.get("restart"): b = b.restart_interval(flags["restart"])
    return b.build()


def _worker_banded(rank, world, port, w, h, ct, ss, q, ret, flags=None):
    dist = _setup(rank, world, port)
    import oracle_lib as O
    import synth
    from pixo_amd import jpeg, sharded
    flags = flags or {}
    px = synth.noise_gray(w, h, 77) if ct == 0 else synth.noise(w, h, 77)
    bpp = 1 if ct == 0 else 3

    def cpu_coeffs(sub, o):  # stands in for the rank's GPU
        return O.coeffs(sub, o.width, o.height, int(o.color_type), int(o.subsampling), o.quality)

    o = _options(w, h, ct, ss, q, flags)
    b = jpeg.band(w, h, ct, ss, world, rank)
    mine = px[b["row_begin"] * w * bpp: b["row_end"] * w * bpp]  # a rank holds only its own rows
    got = sharded.encode_banded(mine, o, coeff_fn=cpu_coeffs)
    if rank == 0:
        ret.put(got == O.encode(px, O.make_options(w, h, ct, q, ss, **flags)))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def _run(target, world, args, timeout=180):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (ret,)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
        assert p.exitcode == 0
    return ret.get(timeout=5)


@pytest.mark.parametrize("case", [(333, 211, 2, 1, 80), (100, 75, 2, 0, 60), (64, 40, 0, 0, 90), (40, 16, 2, 1, 80)])
def test_two_rank_per_band_entropy_and_splice_is_byte_identical(case):
    """(40x16 4:2:0 has ONE MCU row: rank 1's band is empty and must forward nothing.)"""
    assert _run(_worker_banded, 2, case) is True


def _worker_banded_flags(rank, world, port, flags, ret):
    _worker_banded(rank, world, port, 200, 203, 2, 1, 75, ret, flags)


@pytest.mark.parametrize("flags", [{}, {"optimize_huffman": True}])
def test_three_rank_uneven_bands(flags):
    """13 MCU rows over 3 ranks (5 + 4 + 4); with optimised tables the ranks also sum their symbol counts."""
    assert _run(_worker_banded_flags, 3, (flags,)) is True


def _worker_banded_shared(rank, world, port, w, h, ct, ss, q, flags, ret):
    """encode_banded with a SharedFile: every rank writes its body to its final place in one shared-memory file
    (on a GPU node: over its own PCIe link); rank 0 finishes the splice in place."""
    dist = _setup(rank, world, port)
    import numpy as np
    import oracle_lib as O
    import synth
    from pixo_amd import jpeg, sharded
    px = synth.noise(w, h, 77)

    def cpu_coeffs(sub, o):
        return O.coeffs(sub, o.width, o.height, int(o.color_type), int(o.subsampling), o.quality)

    o = _options(w, h, ct, ss, q, flags)
    name = "pixo_test_%d" % port
    size = w * h * 3 + 4096
    shared = sharded.SharedFile(name, size, create=True) if rank == 0 else None
    dist.barrier()
    if rank != 0:
        shared = sharded.SharedFile(name, size, create=False)
    b = jpeg.band(w, h, ct, ss, world, rank)
    mine = px[b["row_begin"] * w * 3: b["row_end"] * w * 3]
    n = sharded.encode_banded(mine, o, coeff_fn=cpu_coeffs, shared=shared)
    if rank == 0:
        ret.put(shared.array()[:n].tobytes() == O.encode(px, O.make_options(w, h, ct, q, ss, **flags)))
    else:
        assert n is None
    dist.barrier()
    shared.close(unlink=rank == 0)
    dist.destroy_process_group()


@pytest.mark.parametrize("flags", [{}, {"optimize_huffman": True}])
def test_three_ranks_write_their_bodies_into_one_shared_file(flags):
    assert _run(_worker_banded_shared, 3, (200, 203, 2, 1, 75, flags)) is True


def test_banded_form_refuses_what_a_band_cannot_code():
    sys.path.insert(0, os.path.dirname(HERE))
    from pixo_amd import jpeg, sharded
    assert not sharded.band_codable(jpeg.JpegOptions.builder(64, 64).progressive(True).build())
    assert not sharded.band_codable(jpeg.JpegOptions.builder(64, 64).restart_interval(3).build())
    assert sharded.band_codable(jpeg.JpegOptions.builder(64, 64).restart_interval(64).build())  # no marker is ever written
    assert sharded.band_codable(jpeg.JpegOptions.builder(64, 64).optimize_huffman(True).build())


def _worker_gathered(rank, world, port, flags, ret):
    dist = _setup(rank, world, port)
    import oracle_lib as O
    import synth
    from pixo_amd import sharded
    w, h, ct, ss, q = 200, 203, 2, 1, 75
    px = synth.noise(w, h, 77)

    def cpu_coeffs(sub, o):
        return O.coeffs(sub, o.width, o.height, int(o.color_type), int(o.subsampling), o.quality)

    got = sharded.encode_gathered(px, _options(w, h, ct, ss, q, flags), coeff_fn=cpu_coeffs)
    if rank == 0:
        ret.put(got == O.encode(px, O.make_options(w, h, ct, q, ss, **flags)))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("flags", [{"restart": 5}, {"progressive": True}, {"progressive": True, "optimize_huffman": True}])
def test_three_rank_gathered_tuple_for_restart_markers_and_progressive_scans(flags):
    assert _run(_worker_gathered, 3, (flags,)) is True


def _worker_gathered_device(rank, world, port, w, h, ct, ss, q, ret):
    """encode_gathered_device with CPU tensors over gloo: equal-sized send buffers, one gather per
    plane, slicing back to the true band sizes, stitched tuple -> entropy stage."""
    dist = _setup(rank, world, port)
    import torch
    import oracle_lib as O
    import synth
    from pixo_amd import jpeg, sharded
    px = synth.noise_gray(w, h, 78) if ct == 0 else synth.noise(w, h, 78)
    bpp = 1 if ct == 0 else 3
    b = jpeg.band(w, h, ct, ss, world, rank)
    mine = torch.from_numpy(px[b["row_begin"] * w * bpp: b["row_end"] * w * bpp].copy())

    def cpu_coeffs(t, o, y, cb, cr):  # stands in for the rank's coefficient kernel
        oy, ocb, ocr = O.coeffs(t.numpy(), o.width, o.height, int(o.color_type), int(o.subsampling), o.quality)
        y[: oy.shape[0]] = torch.from_numpy(oy)
        cb[: ocb.shape[0]] = torch.from_numpy(ocb)
        cr[: ocr.shape[0]] = torch.from_numpy(ocr)

    def host_entropy(y, cb, cr, o):  # stands in for the device entropy stage on rank 0
        return jpeg.entropy_encode(y.numpy(), cb.numpy(), cr.numpy(), o)

    o = _options(w, h, ct, ss, q, {"restart": 3})
    got = sharded.encode_gathered_device(mine, o, coeff_fn=cpu_coeffs, entropy_fn=host_entropy)
    if rank == 0:
        ret.put(got == O.encode(px, O.make_options(w, h, ct, q, ss, restart=3)))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [(333, 211, 2, 1, 80), (100, 72, 2, 0, 60), (64, 40, 0, 0, 90)])
def test_two_rank_device_form_gathers_equal_sized_bands(case):
    assert _run(_worker_gathered_device, 2, case) is True
