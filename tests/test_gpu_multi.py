"""One image across several devices inside one process (`pixo_hip_jpeg_encode_multi`, the band encoder,
the splice) and the lifetime of the per-thread contexts — on the GPU, through the C ABI.  The test box has
ONE GPU: the bands of an image all go to device 0 (a device may be listed more than once), each on its own
host thread with its own context and stream, which exercises every exchange and the splice exactly as 8
devices would."""
import hashlib
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import oracle_lib as O
import synth
from pixo_amd import ColorType, jpeg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _opts(w, h, ct, ss, q, **kw):
    b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss))
    for k, v in kw.items():
        b = getattr(b, k)(v)
    return b.build()


@pytest.mark.parametrize("parts", [1, 2, 3, 8, 13, 30])
@pytest.mark.parametrize("case", [(200, 203, 2, 1, 75), (1027, 333, 2, 1, 80), (97, 161, 2, 0, 90), (64, 100, 0, 0, 50)])
def test_multi_device_file_equals_the_single_device_file(case, parts):
    w, h, ct, ss, q = case
    px = synth.noise_gray(w, h, 5) if ct == 0 else synth.noise(w, h, 5)
    want = O.encode(px, O.make_options(w, h, ct, q, ss))
    assert jpeg.encode_multi(px, _opts(w, h, ct, ss, q), [0] * parts) == want


@pytest.mark.parametrize("parts", [2, 5, 8])
def test_multi_device_optimised_tables_and_smooth_content(parts):
    w, h = 640, 400
    for px in (synth.noise(w, h, 6), synth.gradient_rgb(w, h)):
        for ss in (0, 1):
            want = O.encode(px, O.make_options(w, h, 2, 85, ss, optimize_huffman=True))
            assert jpeg.encode_multi(px, _opts(w, h, 2, ss, 85, optimize_huffman=True), [0] * parts) == want
            assert jpeg.encode_multi(px, _opts(w, h, 2, ss, 85), [0] * parts) == O.encode(px, O.make_options(w, h, 2, 85, ss))


def test_multi_device_bands_of_a_few_bits_and_0xff_at_the_seams():
    w, h = 8, 8 * 24  # flat gray: six bits per block, one block per band
    px = np.full(w * h, 128, np.uint8)
    assert jpeg.encode_multi(px, _opts(w, h, 0, 0, 80), [0] * 24) == O.encode(px, O.make_options(w, h, 0, 80, 0))
    w, h = 48, 16 * 37  # saturated noise at q=100: 0xFF bytes everywhere, also in the bytes two bands share
    px = synth.extremes(w, h, 11)
    for parts in (2, 9, 37):
        assert jpeg.encode_multi(px, _opts(w, h, 2, 1, 100), [0] * parts) == O.encode(px, O.make_options(w, h, 2, 100, 1))


def test_multi_device_falls_back_to_one_device_for_progressive_and_restart_files():
    w, h = 333, 211
    px = synth.noise(w, h, 8)
    for kw, okw in (({"restart_interval": 5}, {"restart": 5}), ({"progressive": True}, {"progressive": True}),
                    ({"progressive": True, "trellis_quant": True, "optimize_huffman": True},
                     {"progressive": True, "trellis": True, "optimize_huffman": True})):
        assert jpeg.encode_multi(px, _opts(w, h, 2, 1, 80, **kw), [0, 0, 0]) == O.encode(px, O.make_options(w, h, 2, 80, 1, **okw))


def test_band_encoder_pieces_equal_the_host_twins():
    """Device band encoder against the host twin, step by step: last DCs, counts, bits and the piece itself
    (head bits, stuffed body, tail bits), at every bit offset modulo 8."""
    w, h, parts = 520, 330, 4
    px = synth.noise(w, h, 12)
    for optimize in (False, True):
        o = _opts(w, h, 2, 1, 80, optimize_huffman=optimize)
        prev = [0, 0, 0]
        for k in range(parts):
            enc = jpeg.BandEncoder(o, parts, k, 0)
            rows = enc.row_end - enc.row_begin
            sub = px[enc.row_begin * w * 3: enc.row_end * w * 3]
            y, cb, cr = O.coeffs(sub, w, rows, 2, 1, 80)
            last = enc.coeffs(sub)
            assert last == [int(y[-1, 0]), int(cb[-1, 0]), int(cr[-1, 0])]
            total = None
            if optimize:
                total = enc.count(prev)
                assert np.array_equal(total, jpeg.band_count_host(y, cb, cr, o, rows, prev))
            bits = enc.lengths(prev, total)
            assert bits == jpeg.band_bits_host(y, cb, cr, o, rows, prev, total)
            for off in range(8):
                assert enc.pack(1000 + off) == jpeg.band_piece_host(y, cb, cr, o, rows, prev, 1000 + off, total), (k, off)
            enc.close()
            prev = last


def test_one_band_encoder_reused_for_two_images_builds_each_image_its_own_tables():
    """A band encoder kept across images (persistent workers do that) must start every pass from scratch: with optimised
    tables the second image's scan has to be coded with the tables of ITS statistics, and `lengths` called twice with
    different statistics must follow the second (the job used to keep `tables_ready` and the first tables)."""
    w, h = 520, 330
    o = _opts(w, h, 2, 1, 80, optimize_huffman=True)
    enc = jpeg.BandEncoder(o, 1, 0, 0)
    try:
        for seed, make in ((12, synth.noise), (3, lambda w_, h_, s_: synth.gradient_rgb(w_, h_)), (5, synth.noise)):
            px = make(w, h, seed)
            y, cb, cr = O.coeffs(px, w, h, 2, 1, 80)
            enc.coeffs(px)
            total = enc.count([0, 0, 0])
            assert np.array_equal(total, jpeg.band_count_host(y, cb, cr, o, h, [0, 0, 0]))
            other = total.copy()
            other[:12] = other[:12][::-1]  # some other statistics first: the second call must win
            other[other == 0] = 1
            enc.lengths([0, 0, 0], other)
            bits = enc.lengths([0, 0, 0], total)
            assert bits == jpeg.band_bits_host(y, cb, cr, o, h, [0, 0, 0], total)
            assert enc.pack(0) == jpeg.band_piece_host(y, cb, cr, o, h, [0, 0, 0], 0, total)
    finally:
        enc.close()


def test_band_encoder_refuses_option_sets_a_band_cannot_code():
    from pixo_amd import error
    with pytest.raises(error.Error, match="baseline scans without restart markers"):
        jpeg.BandEncoder(_opts(64, 64, 2, 1, 80, progressive=True), 2, 0, 0)
    with pytest.raises(error.Error, match="baseline scans without restart markers"):
        jpeg.BandEncoder(_opts(64, 64, 2, 1, 80, restart_interval=3), 2, 0, 0)
    with pytest.raises(error.Error, match="no HIP device"):
        jpeg.BandEncoder(_opts(64, 64, 2, 1, 80), 2, 0, 99)
    with pytest.raises(error.Error, match="no HIP device"):
        jpeg.set_device(99)
    with pytest.raises(error.Error, match="trellis_quant needs the pixels"):
        jpeg.entropy_encode(np.zeros((64, 64), np.int16), np.zeros((64, 64), np.int16), np.zeros((64, 64), np.int16),
                            _opts(64, 64, 2, 0, 80, progressive=True, trellis_quant=True))


def test_config4_16384_image_in_8_bands_matches_the_reference_file():
    """configs[3] the way 8 GPUs do it — eight bands, per-band entropy coding, boundary DCs and bit totals
    exchanged, pieces spliced — with all bands on the one GPU of the test box: the reference's own file
    (178,548,465 bytes, sha256 77cc6cb6...)."""
    w = h = 16384
    px = synth.noise(w, h, 42)
    blob = jpeg.encode_multi(px, _opts(w, h, 2, 1, 80), [0] * 8)
    assert len(blob) == 178548465
    assert hashlib.sha256(blob).hexdigest() == "77cc6cb69a782693c46f2024ac57ebfdfb8411148fa3cef62698c727af36c70c"


def test_device_pixels_are_read_after_their_producer():
    """A device-pointer entry point called right after the kernel that writes its pixels (torch's current
    stream) must see them: the library orders its own stream after the producer stream on the device."""
    import torch
    w = h = 4096
    px = synth.noise(w, h, 42)
    want = hashlib.sha256(jpeg.encode(px, _opts(w, h, 2, 1, 80))).hexdigest()
    src = torch.from_numpy(px).to("cuda:0")
    big = torch.empty(64 << 20, dtype=torch.float32, device="cuda:0")
    jpeg.set_producer_stream(None)  # the NULL stream = torch's default stream
    for _ in range(3):
        d_px = torch.zeros_like(src)
        big.normal_()          # keeps the stream busy for a while ...
        d_px.copy_(src)        # ... before the pixels appear
        assert hashlib.sha256(jpeg.encode_device(d_px, _opts(w, h, 2, 1, 80))).hexdigest() == want
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        jpeg.set_producer_stream(side.cuda_stream)
        d_px = torch.zeros_like(src)
        big.normal_()
        d_px.copy_(src, non_blocking=True)
        assert hashlib.sha256(jpeg.encode_device(d_px, _opts(w, h, 2, 1, 80))).hexdigest() == want
    jpeg.set_producer_stream(None)
    side.synchronize()


_WORKERS = """
    import os, sys, threading
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import synth
    from pixo_amd import jpeg
    w = h = %d
    px = synth.noise(w, h, 42)
    o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420).build()
    sizes = []
    def work():
        for _ in range(%d):
            sizes.append(len(jpeg.encode(px, o)))
"""


def _run_script(body):
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(body)], capture_output=True, text=True, timeout=600)
    err = "\n".join(l for l in r.stderr.splitlines() if "amdgpu.ids" not in l)
    return r.returncode, r.stdout, err


def test_sixteen_threads_ending_at_once_do_not_take_the_process_down():
    """Round 1 crashed here (profiles/r02_thread_exit_crash.txt): 16 threads x a few 4096x4096 encodes, all
    threads end together, the process exits: exit code 0, nothing on stderr."""
    rc, out, err = _run_script((_WORKERS % (ROOT, ROOT, 4096, 3)) + """
    ts = [threading.Thread(target=work) for _ in range(16)]
    for t in ts: t.start()
    for t in ts: t.join()
    print(len(sizes), set(sizes))
    """)
    assert rc == 0 and err == "", (rc, err[-2000:])
    assert out.strip() == "48 {11150133}"


def test_workers_that_outlive_main_and_a_main_thread_that_never_calls_the_library():
    """Daemon-less worker threads are still encoding when the main thread — which never touched the library —
    reaches the end of the script; the interpreter waits for them, then exits: exit code 0, nothing on stderr.
    Second form: main exits with os._exit while workers are mid-call (no destructors at all)."""
    rc, out, err = _run_script((_WORKERS % (ROOT, ROOT, 2048, 6)) + """
    ts = [threading.Thread(target=work) for _ in range(8)]
    for t in ts: t.start()
    print("main done")
    """)
    assert rc == 0 and err == "" and "main done" in out, (rc, err[-2000:])
    rc, out, err = _run_script((_WORKERS % (ROOT, ROOT, 2048, 50)) + """
    import time
    ts = [threading.Thread(target=work, daemon=True) for _ in range(8)]
    for t in ts: t.start()
    time.sleep(1.0)
    print("leaving", len(sizes) > 0, flush=True)
    sys.exit(0)   # daemon threads are abandoned mid-call while the runtime shuts down
    """)
    assert rc == 0 and "leaving True" in out, (rc, err[-2000:])


def test_world_of_one_rccl_batch_and_bench_legs_of_a_multi_gpu_run():
    """What a node with N GPUs runs, on the one GPU the test box has: a world-of-one RCCL group through
    `sharded.encode_batch` (scatter = nothing to send, device arena, size exchange, gather = nothing to receive, one D2H) for
    several option sets, and `bench.py --gpus 1`, whose line must carry the `rccl` block and the configs[3] / configs[2]
    legs a multi-GPU run measures — configs[3]'s file with the reference's sha256."""
    code = textwrap.dedent("""
        import os, sys, json, hashlib
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np, torch, torch.distributed as dist
        import oracle_lib as O, synth
        from pixo_amd import ColorType, jpeg, sharded, error
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29571"
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        for (w, h, ct, ss, n, kw) in [(640, 360, 2, 1, 9, {}), (97, 61, 2, 0, 5, {"optimize_huffman": True}), (64, 48, 0, 0, 3, {}),
                                      (120, 80, 2, 1, 4, {"progressive": True})]:
            imgs = [(synth.noise(w, h, 42 + i) if ct else synth.noise_gray(w, h, 42 + i)) for i in range(n)]
            b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(77).subsampling(jpeg.Subsampling(ss))
            for k, v in kw.items():
                b = getattr(b, k)(v)
            d = torch.from_numpy(np.concatenate(imgs)).cuda()
            arena, offs, lens = sharded.encode_batch(d, b.build(), n)
            for i in range(n):
                assert arena[offs[i]: offs[i] + lens[i]].numpy().tobytes() == O.encode(imgs[i], O.make_options(w, h, ct, 77, ss, **kw)), (w, h, i)
            small = torch.empty(10, dtype=torch.uint8).pin_memory()
            try:
                sharded.encode_batch(d, b.build(), n, out=small)
                raise SystemExit("a 10-byte arena was accepted")
            except error.BufferTooSmall as e:
                assert e.needed == sum(lens)
        # the same batch with the files written into a node-shared arena (every rank over its own PCIe link)
        total = sum(lens)
        for size in (total + 64, total - 1):
            shared = sharded.SharedFile("pixo_gpu_batch_" + str(size), size, create=True)
            try:
                got = sharded.encode_batch(d, b.build(), n, shared=shared)
                assert size >= total
                _, offs2, lens2 = got
                arr = shared.array()
                for i in range(n):
                    assert arr[offs2[i]: offs2[i] + lens2[i]].tobytes() == O.encode(imgs[i], O.make_options(w, h, ct, 77, ss, **kw)), i
            except error.BufferTooSmall as e:
                assert size < total and e.needed == total
            shared.close(unlink=True)
        dist.destroy_process_group()
        print("BATCH_OK")
    """ % (ROOT, ROOT))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "BATCH_OK" in r.stdout, r.stderr[-3000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--blocks", "3",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    assert [l[:1] for l in r.stdout.splitlines() if l.strip()] == ["{"], r.stdout[:2000]  # ONE line on stdout, RCCL's banner included in stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["rccl"]["world"] == 1 and line["rccl"]["backend"] == "nccl" and line["rccl"]["devices"][0]["rank"] == 0
    c4, c3 = line["other_configs"]["c4"], line["other_configs"]["c3_sharded"]
    assert "error" not in c4 and "error" not in c3, (c4, c3)
    assert c4["config"]["file_sha256"] == "77cc6cb69a782693c46f2024ac57ebfdfb8411148fa3cef62698c727af36c70c" and c4["config"]["file_bytes"] == 178548465
    assert c3["config"]["images_per_rank"] == [64] and c3["config"]["file0_sha256"].startswith("d1811ba1761f6b2a")
    assert c4["value"] > 1000 and c3["value"] > 1000
    c4s = line["other_configs"]["c4_shared_arena"]
    assert "error" not in c4s and c4s["config"]["file_sha256"] == c4["config"]["file_sha256"], c4s
    c3s = line["other_configs"]["c3_sharded_shared_arena"]
    assert "error" not in c3s, c3s
    assert c3s["config"]["file0_sha256"] == c3["config"]["file0_sha256"] and c3s["config"]["file_bytes_total"] == c3["config"]["file_bytes_total"]


_RANKS_ON_ONE_GPU = """
import os, sys, hashlib
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
import numpy as np, torch, torch.distributed as dist
import oracle_lib as O, synth
from pixo_amd import ColorType, jpeg, sharded, error
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
ok = True
# --- one image in MCU-row bands, every rank's band RESIDENT ON THE GPU, per-band entropy coding on the GPU ---
for (w, h, ct, ss, q, kw) in [(1000, 700, 2, 1, 80, {}), (333, 517, 2, 0, 91, {}), (640, 200, 0, 0, 60, {}), (512, 512, 2, 1, 75, {"optimize_huffman": True}),
                              (130, 20, 2, 1, 80, {}), (2048, 2048, 2, 1, 85, {}), (777, 1301, 2, 1, 100, {}),
                              (64, 4096, 2, 0, 50, {"optimize_huffman": True}), (4096, 48, 0, 0, 97, {"optimize_huffman": True})]:
    px = synth.noise_gray(w, h, 11) if ct == 0 else (synth.noise(w, h, 11) if w != 2048 else synth.gradient_rgb(w, h))
    b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss))
    for k, v in kw.items():
        b = getattr(b, k)(v)
    opts = b.build()
    band = jpeg.band(w, h, ct, ss, world, rank)
    bpp = 1 if ct == 0 else 3
    mine = torch.from_numpy(px[band["row_begin"] * w * bpp: band["row_end"] * w * bpp].copy()).to(dev)
    want = O.encode(px, O.make_options(w, h, ct, q, ss, **kw))
    for dst in (0, world - 1):
        got = sharded.encode_banded(mine, opts, dst=dst, device=0)
        if rank == dst:
            ok = ok and got == want
        else:
            ok = ok and got is None
    # the same with every rank writing its body into one node-shared, registered segment
    name = "pixo_gpu_ranks_%%s_%%d_%%d" %% (port, w, h)
    size = len(want) + 100
    shared = sharded.SharedFile(name, size, create=True) if rank == 0 else None
    dist.barrier()
    if rank != 0:
        shared = sharded.SharedFile(name, size, create=False)
    shared.register()
    n = sharded.encode_banded(mine, opts, dst=0, device=0, shared=shared)
    if rank == 0:
        ok = ok and n == len(want) and shared.array()[:n].tobytes() == want
    dist.barrier()
    shared.close(unlink=rank == 0)
# --- progressive files cannot be coded in bands: coefficient bands gathered on dst, entropy stage there ---
w, h = 320, 240
px = synth.noise(w, h, 5)
opts = jpeg.JpegOptions.builder(w, h).quality(70).subsampling(jpeg.Subsampling(1)).progressive(True).build()
band = jpeg.band(w, h, 2, 1, world, rank)
mine = torch.from_numpy(px[band["row_begin"] * w * 3: band["row_end"] * w * 3].copy()).to(dev)
got = sharded.encode_gathered_device(mine, opts, dst=0)
if rank == 0:
    ok = ok and got == O.encode(px, O.make_options(w, h, 2, 70, 1, progressive=True))
# --- a batch resident on rank `src`, encoded by every rank on the GPU, files gathered on `dst` ---
for (w, h, n, src, dst, q) in [(640, 360, 7, 0, 0, 82), (96, 64, 2, world - 1, 0, 82), (200, 120, 11, 0, world - 1, 82),
                               (256, 256, 9, 0, 0, 100)]:  # (q = 100 noise: files larger than the first arena a rank reserves — reserve and retry)
    imgs = [synth.noise(w, h, 100 + i) for i in range(n)]
    opts = jpeg.JpegOptions.builder(w, h).quality(q).subsampling(jpeg.Subsampling(1)).build()
    d = torch.from_numpy(np.concatenate(imgs)).to(dev) if rank == src else None
    got = sharded.encode_batch(d, opts, n, src=src, dst=dst, device=0)
    if rank == dst:
        arena, offs, lens = got
        for i in range(n):
            ok = ok and arena[offs[i]: offs[i] + lens[i]].numpy().tobytes() == O.encode(imgs[i], O.make_options(w, h, 2, q, 1))
    else:
        ok = ok and got is None
flag = torch.tensor([1 if ok else 0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
print("RANK", rank, "OK" if ok else "MISMATCH", "ALL", int(flag[0]), "fallbacks", jpeg.lookback_fallbacks(), flush=True)
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 3, 8])
def test_ranks_above_zero_on_a_gpu_bands_batches_and_gathered_tuples(world):
    """N processes, ALL on the one GPU the box has (RCCL refuses that; the exchanges travel over gloo, device tensors staged
    through the host by `sharded._wire`): the first time ranks above 0 run the DEVICE band encoder (non-zero bit offsets, seeded
    predictors, bodies copied into a registered shared segment from a second process), `encode_gathered_device` and
    `encode_batch` on received device tensors.  Every file against the oracle."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    code = _RANKS_ON_ONE_GPU % {"root": ROOT}
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(world), port], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out, err))
    for r, (rc, out, err) in enumerate(outs):
        assert rc == 0, (r, err[-3000:])
        assert "RANK %d OK ALL 1 fallbacks 0" % r in out, (r, out, err[-2000:])


def test_bench_run_of_two_ranks_sharing_the_gpu_measures_every_multi_gpu_leg():
    """`bench.py --gpus 2` as the driver starts it on a node, except that both ranks use GPU 0 and talk over gloo
    (PIXO_BENCH_SHARE_GPU=1: test mode, flagged in the line): every leg a multi-GPU run adds must finish without an error and with the
    reference's / the oracle's bytes — configs[3] in two bands (sha256 of the reference's file), the same through the shared arena,
    configs[2] scattered from rank 0 and gathered, the single-process form — and stdout must carry exactly the one line."""
    import json
    env = dict(os.environ, PIXO_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--blocks", "3",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[:2000]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and "TEST MODE" in line["data"] and line["rccl"]["world"] == 2 and line["rccl"]["backend"] == "gloo"
    assert line["rccl"]["all_reduce_of_rank_plus_1"] == 3 and [d["rank"] for d in line["rccl"]["devices"]] == [0, 1]
    oc = line["other_configs"]
    for name in ("c4", "c4_shared_arena", "c3_sharded", "c3_sharded_shared_arena", "c3_sharded_two_waves", "c3_single_process", "c4_single_process"):
        assert name in oc and "error" not in oc[name], (name, oc.get(name))
    # round 5: every leg says where a step's time goes, rank by rank; the batch also runs in two waves and through the C ABI alone
    assert len(oc["c4"]["phases_ms_by_rank"]) == 2 and "coeffs_ms" in oc["c4"]["phases_ms_by_rank"][1]
    assert len(oc["c3_sharded"]["config"]["phases_ms_by_rank"]) == 2 and "encode_ms" in oc["c3_sharded"]["config"]["phases_ms_by_rank"][1]
    assert oc["c3_sharded_two_waves"]["config"]["waves"] == 2
    assert oc["c3_sharded_two_waves"]["config"]["file_bytes_total"] == oc["c3_sharded"]["config"]["file_bytes_total"] == oc["c3_single_process"]["config"]["file_bytes_total"]
    sha = "77cc6cb69a782693c46f2024ac57ebfdfb8411148fa3cef62698c727af36c70c"
    assert oc["c4"]["config"]["file_sha256"] == sha and oc["c4_shared_arena"]["config"]["file_sha256"] == sha
    assert oc["c4_single_process"]["config"]["file_sha256"] == sha and oc["c4"]["n_gpus"] == 2
    assert oc["c3_sharded"]["config"]["images_per_rank"] == [32, 32] and oc["c3_sharded"]["config"]["file0_sha256"].startswith("d1811ba1761f6b2a")
    assert oc["c3_sharded_shared_arena"]["config"]["file_bytes_total"] == oc["c3_sharded"]["config"]["file_bytes_total"]


# ---- a BATCH over the GPUs of one process (pixo_hip_jpeg_encode_batch_multi; round 5) ----------------------
@pytest.mark.parametrize("parts", [1, 2, 3, 5, 8])
def test_batch_multi_same_device_repeated_equals_the_one_gpu_batch(parts):
    """the same device listed 1-8 times: byte-equal to pixo_hip_jpeg_encode_batch_device_into over the whole batch, device
    pixels and host pixels, pinned and pageable arena"""
    import torch
    w, h, n = 320, 200, 11
    o = _opts(w, h, 2, 1, 80)
    imgs = np.concatenate([synth.noise(w, h, 100 + i) for i in range(n)])
    d = torch.from_numpy(imgs).cuda()
    ref_arena = torch.empty(n * w * h * 2, dtype=torch.uint8).pin_memory()
    ro, rl = jpeg.encode_batch_device_into(ref_arena, d, o, n)
    want = bytes(ref_arena[: ro[-1] + rl[-1]].numpy().tobytes())
    for i in range(n):
        assert want[ro[i]: ro[i] + rl[i]] == O.encode(imgs[i * w * h * 3:(i + 1) * w * h * 3], O.make_options(w, h, 2, 80, 1))
    for src in (d, imgs):  # device memory; host memory
        arena = torch.empty(n * w * h * 2, dtype=torch.uint8).pin_memory()
        offs, lens = jpeg.encode_batch_multi(arena, src, o, n, [0] * parts)
        assert (offs, lens) == (ro, rl)
        assert bytes(arena[: offs[-1] + lens[-1]].numpy().tobytes()) == want
    pageable = np.empty(ro[-1] + rl[-1], np.uint8)
    offs, lens = jpeg.encode_batch_multi(pageable, d, o, n, [0] * parts)
    assert pageable.tobytes() == want


def test_batch_multi_fewer_images_than_devices_size_query_and_small_arena():
    import torch
    from pixo_amd import error
    w, h, n = 200, 120, 3
    o = _opts(w, h, 2, 0, 90)
    imgs = np.concatenate([synth.noise(w, h, 7 + i) for i in range(n)])
    d = torch.from_numpy(imgs).cuda()
    offs, lens = jpeg.encode_batch_multi(None, d, o, n, [0] * 8)  # size query
    need = offs[-1] + lens[-1]
    arena = np.empty(need, np.uint8)
    o2, l2 = jpeg.encode_batch_multi(arena, d, o, n, [0] * 8)
    assert (o2, l2) == (offs, lens)
    for i in range(n):
        assert arena[offs[i]: offs[i] + lens[i]].tobytes() == O.encode(imgs[i * w * h * 3:(i + 1) * w * h * 3], O.make_options(w, h, 2, 90, 0))
    small = np.zeros(need - 1, np.uint8)
    with pytest.raises(error.BufferTooSmall) as e:
        jpeg.encode_batch_multi(small, d, o, n, [0, 0])
    assert e.value.needed == need and not small.any()  # nothing was copied


def test_batch_multi_options_that_are_coded_image_by_image():
    import torch
    w, h, n = 256, 144, 4
    imgs = np.concatenate([synth.noise(w, h, 50 + i) for i in range(n)])
    d = torch.from_numpy(imgs).cuda()
    for kw, okw in (({"optimize_huffman": True}, {"optimize_huffman": True}), ({"progressive": True}, {"progressive": True})):
        o = _opts(w, h, 2, 1, 75, **kw)
        offs, lens = jpeg.encode_batch_multi(None, d, o, n, [0, 0, 0])
        arena = np.empty(offs[-1] + lens[-1], np.uint8)
        jpeg.encode_batch_multi(arena, d, o, n, [0, 0, 0])
        for i in range(n):
            assert arena[offs[i]: offs[i] + lens[i]].tobytes() == O.encode(imgs[i * w * h * 3:(i + 1) * w * h * 3], O.make_options(w, h, 2, 75, 1, **okw))
