"""The N>1 path on CPU: `pixo_amd.sharded.encode_banded` shards the image into MCU-row bands
across ranks (pixo_hip_band), every rank produces its band's coefficients independently (here
with the oracle standing in for the rank's GPU — the point is the sharding and stitching logic,
not the arithmetic), rank 0 gathers them over gloo and runs the product's host entropy coder:
the file must be byte-identical to the single-rank result.  world_size 2, 127.0.0.1 rendezvous."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, ct, ss, q, ret, flags=None):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    import oracle_lib as O
    import synth
    from pixo_amd import ColorType, jpeg
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    px = synth.noise_gray(w, h, 77) if ct == 0 else synth.noise(w, h, 77)
    from pixo_amd import sharded

    def cpu_coeffs(sub, o):  # stands in for the rank's GPU
        return O.coeffs(sub, o.width, o.height, int(o.color_type), int(o.subsampling), o.quality)

    flags = flags or {}
    b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss))
    if flags.get("optimize_huffman"): b = b.optimize_huffman(True)
    if flags.get("progressive"): b = b.progressive(True)
    if flags.get("restart"): b = b.restart_interval(flags["restart"])
    o = b.build()
    got = sharded.encode_banded(px, o, coeff_fn=cpu_coeffs)
    if rank == 0:
        ret.put(got == O.encode(px, O.make_options(w, h, ct, q, ss, **flags)))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [(333, 211, 2, 1, 80), (100, 75, 2, 0, 60), (64, 40, 0, 0, 90), (40, 16, 2, 1, 80)])
def test_two_rank_band_sharding_is_byte_identical(case):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port) + case + (ret,)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def _worker_device_form(rank, world, port, w, h, ct, ss, q, ret):
    """encode_banded_device with CPU tensors over gloo: equal-sized send buffers, one gather per
    plane, slicing back to the true band sizes, stitched tuple -> entropy stage."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    import oracle_lib as O
    import synth
    from pixo_amd import ColorType, jpeg, sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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

    o = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss)).build()
    got = sharded.encode_banded_device(mine, o, coeff_fn=cpu_coeffs, entropy_fn=host_entropy)
    if rank == 0:
        ret.put(got == O.encode(px, O.make_options(w, h, ct, q, ss)))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [(333, 211, 2, 1, 80), (100, 72, 2, 0, 60), (64, 40, 0, 0, 90), (40, 16, 2, 1, 80)])
def test_two_rank_device_form_gathers_equal_sized_bands(case):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_device_form, args=(r, 2, port) + case + (ret,)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


@pytest.mark.parametrize("flags", [{"optimize_huffman": True}, {"restart": 5}, {"progressive": True},
                                   {"progressive": True, "optimize_huffman": True}])
def test_three_rank_bands_with_every_kind_of_file(flags):
    """Uneven bands (3 ranks, 13 MCU rows) and the options that touch the entropy stage: optimised tables,
    restart markers, progressive scans — the stitched tuple gives the single-device file."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    case = (200, 203, 2, 1, 75)
    procs = [ctx.Process(target=_worker, args=(r, 3, port) + case + (ret, flags)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True
