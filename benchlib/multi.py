"""bench.py's multi-GPU legs: configs[3] (one 16384x16384 image over the ranks) and configs[2] (a batch scattered from one GPU), RCCL evidence."""
from .common import *  # noqa: F401,F403
from .kernels import *  # noqa: F401,F403
from . import kernels as _kernels  # noqa: F401


C3_IMAGE0_SHA256 = "d1811ba1761f6b2a76d7f2c3d43418784f38909e0b20af631d5ead73e7d9436a"  # noise(1920,1080,42), made by the reference


MULTI_LEGS_DEADLINE_S = 240.0  # all multi-GPU legs together (they take ~3 s on one GPU); the metric line must not wait longer


# ------------------------------------------------------------------------------------------------------------------
# c4: one 16384x16384 image over the N GPUs
# ------------------------------------------------------------------------------------------------------------------
def ensure_group(job):
    """The exchanges of pixo_amd/sharded.py are torch.distributed calls: a world of one still needs a group."""
    if job.dist is None:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if job.stub:
            dist.init_process_group(backend="gloo", rank=0, world_size=1)
        else:
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=job.dev)
        job.dist = dist


def agree(job, ok):
    """True when EVERY rank says ok (one MIN all_reduce): a rank that failed to prepare an extra workload must not leave the
    others inside that workload's collectives."""
    if job.dist is None:
        return bool(ok)
    t = job.torch.tensor([1 if ok else 0], dtype=job.torch.int64, device=job.wire)
    job.dist.all_reduce(t, op=job.dist.ReduceOp.MIN)
    return bool(t.item())


def rccl_evidence(job):
    """What the process group really was in this run: ranks seen, backend, library version, the device behind every rank,
    and one all_reduce whose result only comes out right when all ranks took part."""
    torch, dist = job.torch, job.dist
    out = {"world": dist.get_world_size(), "backend": dist.get_backend()}
    t = torch.tensor([job.rank + 1], dtype=torch.int64, device=job.wire)
    dist.all_reduce(t)
    out["all_reduce_of_rank_plus_1"] = int(t.item())
    out["all_reduce_expected"] = job.world * (job.world + 1) // 2
    if job.stub:
        mine = {"rank": job.rank, "device": "cpu (stub)", "pid": os.getpid()}
    else:
        pr = torch.cuda.get_device_properties(job.dev)
        mine = {"rank": job.rank, "device": job.gpu_index, "name": pr.name, "pci_bus_id": getattr(pr, "pci_bus_id", None),
                "uuid": str(getattr(pr, "uuid", "")), "pid": os.getpid()}
        try:
            out["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            pass
    seen = [None] * job.world
    dist.all_gather_object(seen, mine)
    out["devices"] = seen
    return out


def gather_phases(job, ph):
    """every rank's per-phase milliseconds of one instrumented call, on rank 0 (a list indexed by rank)"""
    ph = {k: round(v, 3) for k, v in ph.items()}
    if job.dist is None:
        return [ph]
    got = [None] * job.world if job.rank == 0 else None
    job.dist.gather_object(ph, got, dst=0)
    return got


def measure_c4(job, q, steps, warmup, blocks, settle_ms, shared_arena=False):
    """configs[3]: MCU-row bands of ONE 16384x16384 image resident on the N GPUs; a step = the finished file on rank 0.
    Strong scaling: the image is fixed, every rank holds 1/N of it.  Every rank calls; rank 0 gets the result dict.
    shared_arena: the file is assembled in ONE node-shared, registered segment — every rank copies its band's body over its OWN
    PCIe link (1/N of the 178 MB each) instead of all bodies travelling to rank 0 over xGMI and then over rank 0's single link.
    --stub: a 256x192 image through the host twins over gloo (plumbing), the oracle's file as the reference."""
    import synth
    from pixo_amd import jpeg, sharded
    torch = job.torch
    ensure_group(job)
    w, h = (256, 192) if job.stub else (16384, 16384)
    opts = jpeg.JpegOptions.builder(w, h).quality(q).subsampling(jpeg.Subsampling.S420).build()
    b = jpeg.band(w, h, 2, 1, job.world, job.rank)
    rows = b["row_end"] - b["row_begin"]
    mine = synth.noise_rows(w, h, 42, b["row_begin"], b["row_end"])
    state = {}
    shared = None
    if shared_arena:
        name = "pixo_bench_c4_%s" % os.environ.get("MASTER_PORT", "0")
        size = w * h * 3 // 4 + (1 << 20)
        if job.rank == 0:
            shared = sharded.SharedFile(name, size, create=True)
        job.barrier()
        if job.rank != 0:
            shared = sharded.SharedFile(name, size, create=False)
        if not job.stub:
            shared.register()
        state["shared"] = shared
    if job.stub:
        import oracle_lib as O

        def step(i):
            got = sharded.encode_banded(mine, opts, coeff_fn=lambda sub, o: O.coeffs(sub, o.width, o.height, 2, 1, o.quality), shared=shared)
            state["file"] = got if shared is None or got is None else shared.array()[:got].tobytes()
        kev = None
    else:
        d_band = torch.from_numpy(mine).to(job.dev)
        out = torch.empty(w * h * 3 // 4 + (1 << 20), dtype=torch.uint8).pin_memory() if job.rank == 0 and not shared_arena else None

        def step(i):
            state["len"] = sharded.encode_banded(d_band, opts, device=job.gpu_index, out=out, shared=shared)

        # the coefficient kernel of this rank's band alone (roofline object), HIP events on the launch stream
        yb, cbn = jpeg.coefficient_geometry(w, rows, 2, 1)
        ty = torch.empty((yb, 64), dtype=torch.int16, device=job.dev)
        tcb = torch.empty((cbn, 64), dtype=torch.int16, device=job.dev)
        tcr = torch.empty((cbn, 64), dtype=torch.int16, device=job.dev)
        stream = torch.cuda.current_stream().cuda_stream

        def kstep(i):
            jpeg.coefficients_device(d_band, w, rows, 2, 1, q, ty, tcb, tcr, stream=stream)

        job.settle(kstep, settle_ms)
        _, kev = job.time_blocks(kstep, 20, 5, 5)
        del ty, tcb, tcr
    try:
        walls, _ = job.time_blocks(step, steps, warmup, blocks, events=False)
        # one more, instrumented call: where a step's time goes on every rank (diagnosis of the first node run)
        ph = {}
        if job.stub:
            sharded.encode_banded(mine, opts, coeff_fn=lambda sub, o: O.coeffs(sub, o.width, o.height, 2, 1, o.quality), shared=shared, phases=ph)
        else:
            sharded.encode_banded(d_band, opts, device=job.gpu_index, out=out, shared=shared, phases=ph)
        state["phases"] = gather_phases(job, ph)
        if job.rank == 0 and shared is not None and not job.stub:
            out = torch.from_numpy(shared.array()[: state["len"]].copy())
    finally:
        if shared is not None:
            job.barrier()
            shared.close(unlink=job.rank == 0)
    if job.rank != 0:
        return None
    if job.stub:
        blob = state["file"]
        n, digest = len(blob), hashlib.sha256(blob).hexdigest()
        want = hashlib.sha256(O.encode(synth.noise(w, h, 42), O.make_options(w, h, 2, q, 1))).hexdigest()
        if digest != want:
            raise RuntimeError("the banded file differs from the oracle's")
    else:
        n = state["len"]
        digest = hashlib.sha256(out[:n].numpy().tobytes()).hexdigest()
        if (n != 178548465 or digest != C4_SHA256) and not os.environ.get("PIXO_BENCH_ABLATION"):
            raise RuntimeError("the 16384x16384 file differs from the reference's (sha256 %s)" % digest)
    st = block_stats(walls, steps)
    res = {"value": round(w * h / (st["ms_per_step"] * 1e-3) / 1e6, 1), "unit": "Mpixels/s", "n_gpus": job.world, "steps": steps,
           "warmup": warmup, **st, "scaling": "strong",
           "config": {"workload": "configs[3]: single %dx%d RGB8 (noise seed 42) in MCU-row bands across the GPUs, per-band entropy "
                                  "coding, 3 x i16 + u64 exchanged per band over RCCL, %s, spliced on rank 0"
                                  % (w, h, "every band's body copied over its own GPU's PCIe link into one node-shared registered arena" if shared_arena
                                     else "bodies gathered over xGMI"),
                      "width": w, "height": h, "quality": q, "subsampling": "4:2:0", "band_rows_rank0": rows,
                      "file_bytes": int(n), "file_sha256": digest, "sha256_is_the_reference_s": (not job.stub) and digest == C4_SHA256,
                      "parallelism": "one process per GPU, one band per rank"},
           "phases_ms_by_rank": state.get("phases"),
           "roofline": None}
    if kev:
        kernel_ms = statistics.median(kev) / 20
        alg = 6 * w * rows
        achieved = alg / (kernel_ms * 1e-3) / 1e9
        res["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "kernel": "jpeg_coeffs_kernel<M420, L_ALIGNED> on rank 0's band",
                           "algorithmic_bytes_per_launch": alg, "kernel_us_avg": round(kernel_ms * 1e3, 3)}
    return res


def run_c4(job, args):
    steps = max(1, min(args.steps, 20))
    warmup = max(1, min(args.warmup, 3))
    try:
        res = measure_c4(job, args.quality, steps, warmup, max(3, min(args.blocks, 7)), args.settle_ms)
    except RuntimeError as ex:
        raise SystemExit("bench: %s — refusing to report a number" % ex)
    if job.rank != 0:
        job.finish()
        return
    line = {"metric": "Mpixels/s JPEG encode, whole file, one 16384x16384 RGB8 image q=80 4:2:0 across the GPUs (configs[3])",
            "value": res["value"], "unit": "Mpixels/s", "n_gpus": job.world, "steps": steps, "warmup": warmup,
            "ms_per_step": res["ms_per_step"], "ms_per_step_min": res["ms_per_step_min"], "ms_per_step_max": res["ms_per_step_max"],
            "blocks": res["blocks"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "stub" if job.stub else "synthetic", "config": res["config"], "roofline": res["roofline"]}
    job.finish(line)


def measure_c3_sharded(job, q, steps, warmup, blocks, shared_arena=False, waves=1):
    """configs[2] on a node (SURVEY §8e "C3 batch"): 64 x 1920x1080 images RESIDENT ON RANK 0's GPU; a step =
    sharded.encode_batch: whole images to the ranks point to point over xGMI, every rank encodes its share, the files come
    back to rank 0 the same way and cross PCIe once into a pinned arena.  Strong scaling (the batch is fixed).
    --stub: 16 images of 32x24 through the oracle over gloo (plumbing)."""
    import numpy as np
    import synth
    from pixo_amd import jpeg, sharded
    import oracle_lib as O
    torch = job.torch
    ensure_group(job)
    w, h, n = (32, 24, 16) if job.stub else (1920, 1080, 64)
    opts = jpeg.JpegOptions.builder(w, h).quality(q).subsampling(jpeg.Subsampling.S420).build()
    oo = O.make_options(w, h, 2, q, 1)
    d = out = None
    if job.rank == 0:
        host = torch.from_numpy(np.concatenate([synth.noise(w, h, 42 + i) for i in range(n)]))
        d = host if job.stub else host.to(job.dev)
        out = None if job.stub else torch.empty(n * w * h, dtype=torch.uint8).pin_memory()
    px = w * h * 3
    fn = (lambda chunk, o, count: [O.encode(chunk[i * px: (i + 1) * px], oo) for i in range(count)]) if job.stub else None
    state = {}
    shared = None
    if shared_arena:  # one arena in POSIX shared memory that every rank of the node maps: every rank writes ITS files over its own PCIe link
        name = "pixo_bench_%s_%d" % (os.environ.get("MASTER_PORT", "0"), n)
        size = n * px // 2 + (1 << 20)
        if job.rank == 0:
            shared = sharded.SharedFile(name, size, create=True)
        job.barrier()
        if job.rank != 0:
            shared = sharded.SharedFile(name, size, create=False)
        if not job.stub:
            shared.register()

    def step(i):
        state["got"] = sharded.encode_batch(d, opts, n, encode_fn=fn, out=out, device=None if job.stub else job.gpu_index, shared=shared, waves=waves)

    try:
        walls, _ = job.time_blocks(step, steps, warmup, blocks, events=False)
        ph = {}  # one more, instrumented call (the device is synchronised at the step boundaries: not part of the timed blocks)
        sharded.encode_batch(d, opts, n, encode_fn=fn, out=out, device=None if job.stub else job.gpu_index, shared=shared, waves=waves, phases=ph)
        state["phases"] = gather_phases(job, ph)
        if job.rank == 0 and shared is not None:
            _, offs_s, lens_s = state["got"]
            state["got"] = (job.torch.from_numpy(shared.array().copy()), offs_s, lens_s)
    finally:
        if shared is not None:
            job.barrier()
            shared.close(unlink=job.rank == 0)
    if job.rank != 0:
        return None
    arena, offs, lens = state["got"]
    parts = sharded.batch_partition(n, job.world)
    sample = sorted({a for a, b in parts if b > a} | {n - 1})  # the first file of every rank's share + the last file
    for i in sample:
        f = arena[offs[i]: offs[i] + lens[i]].numpy().tobytes()
        if f != O.encode(synth.noise(w, h, 42 + i), oo):
            raise RuntimeError("file %d of the sharded batch differs from the oracle's" % i)
    sha0 = hashlib.sha256(arena[offs[0]: offs[0] + lens[0]].numpy().tobytes()).hexdigest()
    if not job.stub and q == 80 and sha0 != C3_IMAGE0_SHA256:
        raise RuntimeError("file 0 of the sharded batch differs from the reference's")
    st = block_stats(walls, steps)
    return {"value": round(w * h * n / (st["ms_per_step"] * 1e-3) / 1e6, 1), "unit": "Mpixels/s", "n_gpus": job.world, "steps": steps, "warmup": warmup,
            **st, "scaling": "strong",
            "config": {"workload": "configs[2] on a node: %d x %dx%d RGB8 noise (seeds 42..%d) resident on rank 0, q=%d 4:2:0 -> %d files in rank 0's pinned arena"
                                   % (n, w, h, 42 + n - 1, q, n),
                       "waves": waves, "phases_ms_by_rank": state.get("phases"),
                       "images_per_rank": [b - a for a, b in parts], "pixels_scattered_bytes": (n - (parts[0][1] - parts[0][0])) * px,
                       "file_bytes_total": int(sum(lens)), "files_checked_against_oracle": sample, "file0_sha256": sha0,
                       "path": ("sharded.encode_batch(shared=SharedFile): isend/irecv of whole images (one peer per xGMI link) -> "
                                "pixo_hip_jpeg_encode_batch_device_into (device arena) per rank -> all_gather of lengths -> every rank copies its files "
                                "over its OWN PCIe link to their final offsets in one node-shared, registered arena") if shared_arena else
                               ("sharded.encode_batch: isend/irecv of whole images (one peer per xGMI link) -> pixo_hip_jpeg_encode_batch_device_into "
                                "(device arena) per rank -> all_gather of lengths -> isend/irecv of file runs to their final offsets -> one D2H copy")}}


def measure_c4_single_process(job, q, n_dev, steps=3, blocks=3):
    """configs[3] in ONE process (rank 0 only, the other ranks idle): pixo_hip_jpeg_encode_multi drives `n_dev` GPUs from host
    threads — host pixels in over every GPU's own PCIe link, the file's bodies back the same way."""
    import synth
    from pixo_amd import jpeg
    w = h = 16384
    opts = jpeg.JpegOptions.builder(w, h).quality(q).subsampling(jpeg.Subsampling.S420).build()
    px = synth.noise(w, h, 42)
    devices = [0] * n_dev if job.share_gpu else list(range(n_dev))  # (test mode: the bands share GPU 0)
    blob = jpeg.encode_multi(px, opts, devices)  # (also the warm-up: band workers, contexts, pinned buffers)
    digest = hashlib.sha256(blob).hexdigest()
    if len(blob) != 178548465 or digest != C4_SHA256:
        raise RuntimeError("the single-process 16384x16384 file differs from the reference's (sha256 %s)" % digest)
    ts = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(steps):
            jpeg.encode_multi(px, opts, devices)
        ts.append((time.perf_counter() - t0) / steps)
    ts.sort()
    return {"value": round(w * h / ts[len(ts) // 2] / 1e6, 1), "unit": "Mpixels/s", "n_gpus": n_dev, "steps": steps, "blocks": blocks,
            "ms_per_step": round(ts[len(ts) // 2] * 1e3, 3), "ms_per_step_min": round(ts[0] * 1e3, 3), "ms_per_step_max": round(ts[-1] * 1e3, 3),
            "scaling": "strong",
            "config": {"workload": "configs[3], single process: pixo_hip_jpeg_encode_multi over devices %s; 805 MB of HOST pixels in over PCIe, "
                                   "178.5 MB file out as Python bytes" % devices, "file_bytes": len(blob), "file_sha256": digest,
                       "sha256_is_the_reference_s": True}}


def measure_c3_single_process(job, q, n_dev, steps=5, blocks=3):
    """configs[2] in ONE process (rank 0 only): pixo_hip_jpeg_encode_batch_multi — the 64 x 1080p images resident on GPU 0, the other
    GPUs' shares by peer copies (one peer per xGMI link), every GPU encodes its share and copies its files over its OWN PCIe link
    to their final place in one pinned arena.  The torch-free form of c3_sharded_shared_arena."""
    import numpy as np
    import synth
    from pixo_amd import jpeg
    import oracle_lib as O
    torch = job.torch
    w, h, n = 1920, 1080, 64
    opts = jpeg.JpegOptions.builder(w, h).quality(q).subsampling(jpeg.Subsampling.S420).build()
    d = torch.from_numpy(np.concatenate([synth.noise(w, h, 42 + i) for i in range(n)])).to(job.dev)
    arena = torch.empty(n * w * h, dtype=torch.uint8).pin_memory()
    devices = [0] * n_dev if job.share_gpu else list(range(n_dev))
    job.sync()
    offs, lens = jpeg.encode_batch_multi(arena, d, opts, n, devices)  # (also the warm-up: workers, contexts, device buffers)
    for i in sorted({n * k // n_dev for k in range(n_dev)} | {n - 1}):
        if arena[offs[i]: offs[i] + lens[i]].numpy().tobytes() != O.encode(synth.noise(w, h, 42 + i), O.make_options(w, h, 2, q, 1)):
            raise RuntimeError("file %d of the single-process batch differs from the oracle's" % i)
    ts = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(steps):
            jpeg.encode_batch_multi(arena, d, opts, n, devices)
        ts.append((time.perf_counter() - t0) / steps)
    ts.sort()
    del d, arena
    return {"value": round(w * h * n / ts[len(ts) // 2] / 1e6, 1), "unit": "Mpixels/s", "n_gpus": n_dev, "steps": steps, "blocks": blocks,
            "ms_per_step": round(ts[len(ts) // 2] * 1e3, 3), "ms_per_step_min": round(ts[0] * 1e3, 3), "ms_per_step_max": round(ts[-1] * 1e3, 3),
            "scaling": "strong",
            "config": {"workload": "configs[2], single process: pixo_hip_jpeg_encode_batch_multi over devices %s; 64 x 1920x1080 RGB8 noise resident on "
                                   "device %d, files into one pinned arena" % (devices, devices[0]), "file_bytes_total": int(sum(lens))}}


def guarded_multi_gpu_extras(job, args):
    """multi_gpu_extras + the final barrier on a worker thread with a DEADLINE.  These legs run collectives that no
    single-GPU box of this project's sessions could ever exercise with N > 1 ranks; if one of them hangs on a real node, the
    run must still print its metric line.  Returns (results or None, timed_out).  After a timeout the process group is in an
    unknown state: the caller prints its line and leaves with os._exit (no barrier, no destroy)."""
    import threading
    box = {}

    def work():
        try:
            if not job.stub:
                job.torch.cuda.set_device(job.gpu_index)  # (the current device is per thread)
            box["out"] = multi_gpu_extras(job, args)
            if job.dist is not None:
                job.dist.barrier()
            box["done"] = True
        except BaseException as ex:
            box["error"] = repr(ex)

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(MULTI_LEGS_DEADLINE_S)
    if th.is_alive() or not box.get("done"):
        out = box.get("out") or {}
        out["multi_gpu_legs"] = {"error": box.get("error") or "no result within %.0f s: abandoned" % MULTI_LEGS_DEADLINE_S}
        return out, True
    return box["out"], False


def leave_without_teardown(line):
    """After a multi-GPU leg was abandoned: print the line (rank 0) and end the process at once — collectives may be stuck."""
    if line is not None:
        emit(line)
    sys.stderr.flush()
    os._exit(0)


def multi_gpu_extras(job, args):
    """Every rank calls (collectives inside).  configs[3] and configs[2] over the ranks of THIS run + what the process group
    was.  Each leg under try/except and behind an `agree` round; the metric line does not depend on them."""
    out = {}
    try:
        ensure_group(job)
        out["rccl"] = rccl_evidence(job)
    except BaseException as ex:
        out["rccl"] = {"error": repr(ex)}
    small = job.stub
    legs = (("c4", lambda: measure_c4(job, args.quality, 2 if small else 5, 1, 3, 0 if small else QUICK_SETTLE_MS)),
            ("c4_shared_arena", lambda: measure_c4(job, args.quality, 2 if small else 5, 1, 3, 0, shared_arena=True)),
            ("c3_sharded", lambda: measure_c3_sharded(job, args.quality, 2 if small else 5, 1, 3)),
            ("c3_sharded_shared_arena", lambda: measure_c3_sharded(job, args.quality, 2 if small else 5, 1, 3, shared_arena=True)))
    if job.world > 1:  # the batch in two waves: the second half of every share travels while the first half is encoded
        legs += (("c3_sharded_two_waves", lambda: measure_c3_sharded(job, args.quality, 2 if small else 5, 1, 3, waves=2)),)
    if not job.stub:  # (rank 0 alone; the others wait in the next `agree`)
        legs += (("c3_single_process", lambda: measure_c3_single_process(job, args.quality, job.world) if job.rank == 0 else None),)
    if job.world > 1 and not job.stub:
        legs += (("c4_single_process", lambda: measure_c4_single_process(job, args.quality, job.world) if job.rank == 0 else None),)
    for name, fn in legs:
        t0 = time.perf_counter()
        if not agree(job, True):
            out[name] = {"error": "a rank could not start this leg"}
            continue
        try:
            res = fn()
            ok = True
        except BaseException as ex:  # (a rank-local failure after the collectives: the others have finished the leg)
            res, ok = {"error": repr(ex)}, False
        if job.rank == 0:
            if isinstance(res, dict):
                res["leg_wall_s"] = round(time.perf_counter() - t0, 2)
            out[name] = res
        if not job.stub:
            job.torch.cuda.empty_cache()
    return out


def run_c4_single_process(job, args):
    """configs[3] in ONE process: pixo_hip_jpeg_encode_multi spreads the 16384x16384 image's MCU-row bands over the N GPUs — a
    persistent host thread per band, the band's rows over that GPU's own PCIe link, per-band entropy coding, three tiny
    exchanges through shared memory, every body copied to its final place in the file.  The pixels start in HOST memory (this
    entry's contract), so a step includes their way over PCIe: strong scaling over N links."""
    import synth
    from pixo_amd import jpeg
    torch = job.torch
    w = h = 16384
    n = args.gpus
    have = torch.cuda.device_count()
    devices = [i % max(have, 1) for i in range(n)]
    opts = jpeg.JpegOptions.builder(w, h).quality(args.quality).subsampling(jpeg.Subsampling.S420).build()
    px = synth.noise(w, h, 42)
    state = {}

    def step(i):
        state["file"] = jpeg.encode_multi(px, opts, devices)

    steps = max(1, min(args.steps, 5))
    walls, _ = job.time_blocks(step, steps, 1, max(3, min(args.blocks, 5)), events=False)
    blob = state["file"]
    digest = hashlib.sha256(blob).hexdigest()
    if (len(blob) != 178548465 or digest != C4_SHA256) and not os.environ.get("PIXO_BENCH_ABLATION"):
        raise SystemExit("bench: the 16384x16384 file differs from the reference's — refusing to report a number")
    st = block_stats(walls, steps)
    line = {"metric": "Mpixels/s JPEG encode, whole file from host pixels, one 16384x16384 RGB8 image q=80 4:2:0 across the GPUs of one process (configs[3])",
            "value": round(w * h / (st["ms_per_step"] * 1e-3) / 1e6, 1), "unit": "Mpixels/s", "n_gpus": n, "steps": steps, "warmup": 1,
            "ms_per_step": st["ms_per_step"], "ms_per_step_min": st["ms_per_step_min"], "ms_per_step_max": st["ms_per_step_max"],
            "blocks": st["blocks"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[3], single process: pixo_hip_jpeg_encode_multi over %d band(s) on device(s) %s; 805 MB of pixels in "
                                   "over PCIe, 178.5 MB file out (as Python bytes: one more copy)" % (n, sorted(set(devices))),
                       "width": w, "height": h, "quality": args.quality, "subsampling": "4:2:0", "file_bytes": len(blob), "file_sha256": digest,
                       "devices_visible": have, "parallelism": "one process, one persistent host thread per band"},
            "roofline": None}
    job.finish(line)


__all__ = [n for n in dir() if not n.startswith("__")]
