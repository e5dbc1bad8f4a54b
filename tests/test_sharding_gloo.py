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
    for attempt in range(3):
        ret = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=target, args=(r, world, port) + args + (ret,)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout)
        if all(p.exitcode == 0 for p in procs):
            return ret.get(timeout=5)
        # (the port found free above can be taken by another test process before rank 0 binds it — pytest -n: every rank
        # then dies within its rendezvous; anything else fails again and is reported)
        for p in procs:
            if p.is_alive():
                p.kill()
                p.join(10)
        if attempt == 2:
            assert [p.exitcode for p in procs] == [0] * world
    return None


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


# ----------------------------------------------------------------------------------------------------------------------
# round 4: 8 ranks (the node's shape), the C4 partition rule scaled down, the shared segment's bound, the batch form
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [
    (256, 128 * 16 // 8, 2, 1, 80),   # 16 MCU rows over 8 ranks = 2 each: configs[3]'s rule (1024 MCU rows -> 128 per rank) scaled down
    (100, 16 * 13, 2, 1, 70),         # 13 MCU rows over 8 ranks: 2,2,2,2,2,1,1,1 (uneven)
    (64, 16 * 5 - 3, 2, 1, 85),       # 5 MCU rows over 8 ranks: ranks 5, 6, 7 hold nothing and forward their predecessor's DCs
    (72, 8 * 11, 2, 0, 60),           # 4:4:4: 11 block rows
    (40, 8 * 3, 0, 0, 90),            # gray, 3 block rows over 8 ranks
])
def test_eight_rank_bands_uneven_and_empty(case):
    assert _run(_worker_banded, 8, case, timeout=300) is True


def test_eight_rank_bands_with_summed_symbol_counts():
    assert _run(_worker_banded_flags, 8, ({"optimize_huffman": True},), timeout=300) is True


def test_eight_rank_partition_rule_of_config_4():
    """jpeg.band over 8 parts: contiguous MCU-row bands that cover the image once, sizes differ by at most one MCU row;
    16384 rows of 4:2:0 -> 128 MCU rows = 2048 pixel rows per rank."""
    sys.path.insert(0, os.path.dirname(HERE))
    from pixo_amd import jpeg
    bands = [jpeg.band(16384, 16384, 2, 1, 8, r) for r in range(8)]
    assert [(b["row_begin"], b["row_end"]) for b in bands] == [(2048 * r, 2048 * (r + 1)) for r in range(8)]
    for (w, h, ct, ss) in [(100, 16 * 13, 2, 1), (64, 77, 2, 1), (40, 24, 0, 0), (33, 1000, 2, 0)]:
        bands = [jpeg.band(w, h, ct, ss, 8, r) for r in range(8)]
        unit = 16 if (ct and ss) else 8
        assert bands[0]["row_begin"] == 0 and bands[-1]["row_end"] == h
        assert all(a["row_end"] == b["row_begin"] for a, b in zip(bands, bands[1:]))
        sizes = [-(-(b["row_end"] - b["row_begin"]) // unit) for b in bands]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def _worker_shared_too_small(rank, world, port, ret):
    """a SharedFile smaller than the file must make EVERY rank raise BufferTooSmall (with the size needed)
    before any byte moves — nobody writes past the mapping, nobody is left in the barrier."""
    dist = _setup(rank, world, port)
    import oracle_lib as O
    import synth
    from pixo_amd import error, jpeg, sharded
    w, h, ct, ss, q = 200, 203, 2, 1, 75
    px = synth.noise(w, h, 77)

    def cpu_coeffs(sub, o):
        return O.coeffs(sub, o.width, o.height, int(o.color_type), int(o.subsampling), o.quality)

    o = _options(w, h, ct, ss, q, {})
    want = O.encode(px, O.make_options(w, h, ct, q, ss))
    name = "pixo_small_%d" % port
    size = len(want) - 100
    shared = sharded.SharedFile(name, size, create=True) if rank == 0 else None
    dist.barrier()
    if rank != 0:
        shared = sharded.SharedFile(name, size, create=False)
    shared.array()[:] = 0xAB
    b = jpeg.band(w, h, ct, ss, world, rank)
    mine = px[b["row_begin"] * w * 3: b["row_end"] * w * 3]
    try:
        sharded.encode_banded(mine, o, coeff_fn=cpu_coeffs, shared=shared)
        ok = False
    except error.BufferTooSmall as e:
        ok = e.needed == len(want)
    dist.barrier()  # (every rank got here: nobody hangs)
    ok = ok and bool((shared.array() == 0xAB).all())
    assert sharded.shared_file_bound(o) >= len(want)
    if rank == 0:
        ret.put(ok)
    else:
        assert ok
    dist.barrier()
    shared.close(unlink=rank == 0)
    dist.destroy_process_group()


def test_shared_file_too_small_raises_on_every_rank_and_writes_nothing():
    assert _run(_worker_shared_too_small, 3, ()) is True


def _worker_batch(rank, world, port, n, w, h, ct, ss, q, src, dst, flags, small_out, ret):
    """sharded.encode_batch over gloo with CPU tensors: the scatter of whole images from `src`, the size exchange and the
    gather of the files on `dst` are the calls a GPU node makes; the oracle stands in for every rank's encoder."""
    dist = _setup(rank, world, port)
    import numpy as np
    import torch
    import oracle_lib as O
    import synth
    from pixo_amd import error, sharded
    bpp = 1 if ct == 0 else 3
    px = w * h * bpp
    images = [synth.noise_gray(w, h, 42 + i) if ct == 0 else synth.noise(w, h, 42 + i) for i in range(n)]
    flags = dict(flags)
    waves, fail_rank = flags.pop("_waves", 1), flags.pop("_fail_rank", None)  # (test controls, not encoder options)
    oo = O.make_options(w, h, ct, q, ss, **flags)

    def cpu_encode(chunk, o, count):
        assert chunk.size == count * px
        return [O.encode(chunk[i * px: (i + 1) * px], oo) for i in range(count)]

    o = _options(w, h, ct, ss, q, flags)
    batch = torch.from_numpy(np.concatenate(images)) if rank == src else None  # only src holds pixels
    lo, hi = sharded.batch_partition(n, world)[rank]
    if small_out:
        out = torch.empty(64, dtype=torch.uint8) if rank == dst else None
        try:  # (round 5: dst's capacity travels with the sizes — EVERY rank raises, before any file moves)
            sharded.encode_batch(batch, o, n, src=src, dst=dst, encode_fn=cpu_encode, out=out)
            ok = False
        except error.BufferTooSmall as e:
            ok = e.needed == sum(len(O.encode(im, oo)) for im in images)
    else:
        phases = {}
        if fail_rank is not None:  # one rank's encode step throws: every rank must raise, nobody may hang in the exchange
            def failing(chunk, o_, count):
                if rank == fail_rank:
                    raise ValueError("boom on rank %d" % rank)
                return cpu_encode(chunk, o_, count)
            try:
                sharded.encode_batch(batch, o, n, src=src, dst=dst, encode_fn=failing, waves=waves)
                ok = False
            except ValueError:
                ok = rank == fail_rank
            except RuntimeError as e:
                ok = rank != fail_rank and str(fail_rank) in str(e)
            dist.barrier()
            if rank == dst:
                ret.put(bool(ok))
            else:
                assert ok
            dist.destroy_process_group()
            return
        got = sharded.encode_batch(batch, o, n, src=src, dst=dst, encode_fn=cpu_encode, waves=waves, phases=phases)
        assert "total_ms" in phases and "sizes_ms" in phases and ("encode_ms" in phases or hi == lo)
        if rank == dst:
            arena, offs, lens = got
            ok = len(offs) == n and offs[0] == 0 and all(offs[i + 1] == offs[i] + lens[i] for i in range(n - 1))
            for i in range(n):
                ok = ok and arena[offs[i]: offs[i] + lens[i]].numpy().tobytes() == O.encode(images[i], oo)
        else:
            ok = got is None
    dist.barrier()
    if rank == dst:
        ret.put(bool(ok))
    else:
        assert ok
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,case", [
    (2, 5, (48, 40, 2, 1, 80, 0, 0, {})),                            # 3 + 2 images
    (3, 7, (40, 24, 2, 0, 60, 0, 0, {"optimize_huffman": True})),    # 3 + 2 + 2, per-image tables
    (3, 4, (33, 17, 0, 0, 90, 1, 2, {})),                            # gray; src = 1, dst = 2: the files end up on a rank that held no pixels
    (8, 64, (32, 24, 2, 1, 80, 0, 0, {})),                           # configs[2]'s shape on a node: 64 images, 8 per rank
    (8, 5, (32, 16, 2, 1, 75, 3, 0, {})),                            # fewer images than ranks: ranks 5-7 neither receive nor send
    (8, 13, (24, 24, 2, 0, 50, 7, 7, {"progressive": True})),        # uneven 2,2,2,2,2,1,1,1
])
def test_batch_scattered_from_one_rank_and_files_gathered(world, n, case):
    assert _run(_worker_batch, world, (n,) + case + (False,), timeout=300) is True


def test_batch_output_too_small_raises_on_every_rank_before_any_file_moves():
    assert _run(_worker_batch, 3, (5, 48, 40, 2, 1, 80, 0, 1, {}, True)) is True


@pytest.mark.parametrize("world,n,case", [
    (2, 5, (48, 40, 2, 1, 80, 0, 0, {"_waves": 2})),
    (3, 13, (40, 24, 2, 0, 60, 1, 2, {"_waves": 2})),       # shares of 5, 4, 4 in parts of 3 + 2, 2 + 2, 2 + 2; src != dst
    (8, 64, (32, 24, 2, 1, 80, 0, 0, {"_waves": 2})),       # configs[2]'s shape on a node, two waves of 4 images per rank
    (8, 9, (32, 16, 2, 1, 75, 3, 0, {"_waves": 2})),        # fewer than 2 images per rank: one wave after all
])
def test_batch_in_two_waves_gives_the_same_files(world, n, case):
    assert _run(_worker_batch, world, (n,) + case + (False,), timeout=300) is True


@pytest.mark.parametrize("world,fail_rank", [(3, 1), (3, 0), (8, 5)])
def test_batch_a_failing_rank_makes_every_rank_raise(world, fail_rank):
    assert _run(_worker_batch, world, (2 * world + 1, 32, 24, 2, 1, 80, 0, 0, {"_fail_rank": fail_rank}, False), timeout=300) is True


def test_batch_partition_rule():
    sys.path.insert(0, os.path.dirname(HERE))
    from pixo_amd import sharded
    assert sharded.batch_partition(64, 8) == [(8 * r, 8 * r + 8) for r in range(8)]
    assert sharded.batch_partition(5, 8) == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 5), (5, 5), (5, 5)]
    assert sharded.batch_partition(13, 8) == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 10), (10, 11), (11, 12), (12, 13)]
    assert sharded.batch_partition(0, 3) == [(0, 0)] * 3


def _worker_batch_shared(rank, world, port, n, w, h, ret):
    """encode_batch with a SharedFile: every rank writes its run of files to its final place in one shared arena."""
    dist = _setup(rank, world, port)
    import numpy as np
    import torch
    import oracle_lib as O
    import synth
    from pixo_amd import error, sharded
    px = w * h * 3
    images = [synth.noise(w, h, 42 + i) for i in range(n)]
    oo = O.make_options(w, h, 2, 80, 1)
    want = [O.encode(im, oo) for im in images]

    def cpu_encode(chunk, o, count):
        return [O.encode(chunk[i * px: (i + 1) * px], oo) for i in range(count)]

    o = _options(w, h, 2, 1, 80, {})
    batch = torch.from_numpy(np.concatenate(images)) if rank == 0 else None
    ok = True
    for size in (sum(len(f) for f in want) + 100, sum(len(f) for f in want) - 1):  # large enough; one byte short
        name = "pixo_batch_%d_%d" % (port, size)
        shared = sharded.SharedFile(name, size, create=True) if rank == 0 else None
        dist.barrier()
        if rank != 0:
            shared = sharded.SharedFile(name, size, create=False)
        try:
            got = sharded.encode_batch(batch, o, n, encode_fn=cpu_encode, shared=shared)
            fits = True
        except error.BufferTooSmall as e:
            fits = False
            ok = ok and e.needed == sum(len(f) for f in want)
        ok = ok and fits == (size >= sum(len(f) for f in want))
        if fits and rank == 0:
            _, offs, lens = got
            arr = shared.array()
            ok = ok and all(arr[offs[i]: offs[i] + lens[i]].tobytes() == want[i] for i in range(n))
        dist.barrier()
        shared.close(unlink=rank == 0)
    if rank == 0:
        ret.put(bool(ok))
    else:
        assert ok
    dist.barrier()
    dist.destroy_process_group()


def test_batch_files_written_by_every_rank_into_one_shared_arena():
    assert _run(_worker_batch_shared, 3, (7, 40, 24), timeout=300) is True


def test_shared_file_replaces_a_stale_segment_of_the_same_name():
    from pixo_amd import sharded
    name = "pixo_test_stale_%d" % os.getpid()
    a = sharded.SharedFile(name, 4096, create=True)
    a.array()[:4] = 7
    a.shm.close()  # (the run "died": mapping gone, name left behind)
    b = sharded.SharedFile(name, 8192, create=True)
    try:
        assert b.array().size == 8192 and int(b.array()[:4].sum()) == 0
    finally:
        b.close(unlink=True)


def test_shared_file_refuses_a_name_that_a_live_process_holds():
    """create=True must not unlink a segment that another LIVE job uses; a dead maker's segment is replaced."""
    import subprocess
    from pixo_amd import sharded
    name = "pixo_test_live_%d" % os.getpid()
    code = ("import sys, time; sys.path.insert(0, %r); from pixo_amd import sharded; s = sharded.SharedFile(%r, 4096, create=True); "
            "s.array()[:4] = 9; print('up', flush=True); time.sleep(30)" % (os.path.dirname(HERE), name))
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)
    try:
        assert p.stdout.readline().strip() == "up"
        with pytest.raises(FileExistsError):
            sharded.SharedFile(name, 4096, create=True)
        peer = sharded.SharedFile(name, 4096, create=False)  # attaching is what the other ranks of that job do: fine
        assert int(peer.array()[:4].sum()) == 36
        peer.close()
    finally:
        p.kill()
        p.wait()
    b = sharded.SharedFile(name, 8192, create=True)  # its maker is dead now: stale, replaced
    try:
        assert b.array().size == 8192 and int(b.array()[:4].sum()) == 0
    finally:
        b.close(unlink=True)


def test_shared_file_header_judges_the_maker_by_pid_start_time_and_namespace():
    """the maker's identity lives in the segment's own header.  A header naming this very pid with ANOTHER start time
    (a reused pid) is a dead maker: replaced.  A maker in another PID namespace cannot be judged: refused.  No header at all (an
    older version's segment): replaced."""
    from multiprocessing import shared_memory
    from pixo_amd import sharded
    SF = sharded.SharedFile
    name = "pixo_test_hdr_%d" % os.getpid()

    def raw(fill):
        seg = shared_memory.SharedMemory(name=name, create=True, size=4096 + SF._HEADER)
        fill(seg.buf)
        seg.close()
        SF._untrack(seg)

    def header(buf, pid, start, ns):
        buf[8:32] = pid.to_bytes(8, "little") + start.to_bytes(8, "little") + ns.to_bytes(8, "little")
        buf[:8] = SF._MAGIC
    # a live pid (ours' parent, say pid 1 is always there) whose start time does not match: the pid was reused -> stale
    raw(lambda b: header(b, 1, SF._start_time(1) + 12345, SF._pid_namespace()))
    a = SF(name, 4096, create=True)
    a.close(unlink=True)
    # another PID namespace: never unlinked
    raw(lambda b: header(b, 1, 1, SF._pid_namespace() + 1))
    with pytest.raises(FileExistsError):
        SF(name, 4096, create=True)
    shared_memory.SharedMemory(name=name, create=False).unlink()
    # garbage instead of a header: nobody's -> replaced
    raw(lambda b: b.__setitem__(slice(0, 8), b"whatever"))
    c = SF(name, 4096, create=True)
    assert c.array().size == 4096
    c.close(unlink=True)
