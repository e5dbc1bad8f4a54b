"""bench.py's metric runs: --workload c2 / c2_444 / c3 / c1 / c2_unaligned (run_coeffs) and c5 (run_png)."""
from .common import *  # noqa: F401,F403
from .cpu import *  # noqa: F401,F403
from . import cpu as _cpu  # noqa: F401
from .roofline import *  # noqa: F401,F403
from . import roofline as _roofline  # noqa: F401
from .kernels import *  # noqa: F401,F403
from . import kernels as _kernels  # noqa: F401
from .files import *  # noqa: F401,F403
from . import files as _files  # noqa: F401
from .multi import *  # noqa: F401,F403
from . import multi as _multi  # noqa: F401


def run_coeffs(job, args):
    wl = CoeffWorkload(job, args.workload, args.quality)
    settled = job.settle(wl.step, 0 if job.stub else args.settle_ms)
    walls, evs = job.time_blocks(wl.step, args.steps, args.warmup, args.blocks)
    if job.rank == 0 and job.world == 1:
        measure_engine_clock(job)
    copy_ms = None
    if job.rank == 0 and not job.stub and job.world == 1 and wl.batch == 1:
        # right behind the metric's blocks, same clocks, same protocol: the plain copy of the kernel's bytes; then the
        # kernel once more, so that the pair (kernel, copy) is also available in the order copy -> kernel
        try:
            cstep = wl.copy_step_factory()
            if cstep is not None:
                _, cevs = job.time_blocks(cstep, args.steps, min(args.warmup, 20), args.blocks)
                copy_ms = statistics.median(cevs) / args.steps
                _, kevs2 = job.time_blocks(wl.step, args.steps, min(args.warmup, 20), max(3, args.blocks // 3))
                kernel_after_copy_ms = statistics.median(kevs2) / args.steps
                del cstep
        except BaseException as ex:  # the metric must not depend on the comparison
            copy_ms = None
            sys.stderr.write("bench: same-run copy failed: %r\n" % (ex,))
    if job.rank == 0 and not job.stub and not os.environ.get("PIXO_BENCH_ABLATION"):
        wl.check()
    multi = (not args.no_extras) and args.workload == "c2" and not os.environ.get("PIXO_BENCH_ABLATION")
    if job.rank != 0:
        if multi:
            del wl
            if not job.stub:
                job.torch.cuda.empty_cache()
            _, abandoned = guarded_multi_gpu_extras(job, args)
            if abandoned:
                leave_without_teardown(None)
        job.finish()
        return
    st = block_stats(walls, args.steps)
    pixels_per_step = wl.w * wl.h * wl.batch
    value = pixels_per_step * job.world / (st["ms_per_step"] * 1e-3) / 1e6
    kernel_ms = (statistics.median(evs) / args.steps) if evs else st["ms_per_step"]
    line = {
        "metric": "Mpixels/s JPEG encode (RGB→YCbCr→DCT→quant), 4096×4096 q=80" if args.workload == "c2"
                  else "Mpixels/s JPEG encode (RGB→YCbCr→DCT→quant), %s" % args.workload,
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": job.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": st["ms_per_step"], "ms_per_step_min": st["ms_per_step_min"], "ms_per_step_max": st["ms_per_step_max"],
        "blocks": st["blocks"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "stub" if job.stub else ("synthetic — TEST MODE: the %d ranks share ONE GPU over gloo (PIXO_BENCH_SHARE_GPU), not an N-GPU measurement" % job.world
                                                           if job.share_gpu else "synthetic"),
        "config": {"workload": wl.label, "width": wl.w, "height": wl.h, "batch": wl.batch, "quality": wl.q,
                   "subsampling": "4:2:0" if wl.ss else "4:4:4", "buffers_rotated": wl.nbuf,
                   "working_set_MiB": round(wl.nbuf * (wl.in_bytes + wl.out_bytes) / 2**20, 1),
                   "settle_launches_before_warmup": settled,
                   "timing": "median of %d blocks of %d steps, each block barrier+synchronize bracketed, max over ranks" % (st["blocks"], args.steps),
                   "parallelism": "one process per GPU, images sharded across ranks, no collective"},
        "roofline": wl.roofline(kernel_ms, copy_ms),
    }
    if copy_ms:
        line["roofline"]["kernel_us_after_copy"] = round(kernel_after_copy_ms * 1e3, 3)
    if evs:
        per = sorted(e / args.steps * 1e3 for e in evs)
        line["roofline"]["kernel_us_block_min"], line["roofline"]["kernel_us_block_max"] = round(per[0], 3), round(per[-1], 3)
    extras = (not args.no_extras) and job.world == 1 and not job.stub and not os.environ.get("PIXO_BENCH_ABLATION")
    if extras and wl.batch == 1 and args.workload in ("c2", "c2_444"):
        line["whole_file"] = whole_file(job, wl)
        line["small_files"] = small_files(wl)
    if extras and args.workload == "c2":
        others = {}
        del wl.ins, wl.outs
        job.torch.cuda.empty_cache()
        # c2_same_state: the metric's own workload measured again by the same short protocol, right between the others, so
        # that ratios such as unaligned / aligned compare like with like (the ratios must be same-state)
        for name in ("c3", "c2_444", "c2", "c2_unaligned"):
            try:
                others["c2_same_state" if name == "c2" else name] = quick_kernel(job, name, args.quality)
            except BaseException as ex:  # the metric line must not depend on these
                others[name] = {"error": repr(ex)}
        try:
            others["c2_unaligned"]["over_c2_same_state"] = round(others["c2_unaligned"]["kernel_us"] / others["c2_same_state"]["kernel_us"], 3)
        except Exception:
            pass
        try:
            others["c1"] = config_1(job, args.quality)
        except BaseException as ex:
            others["c1"] = {"error": repr(ex)}
        try:
            others["c3_whole_file"] = batch_whole_files(job, args.quality)
        except BaseException as ex:
            others["c3_whole_file"] = {"error": repr(ex)}
        try:
            others["c5"] = quick_png(job)
        except BaseException as ex:
            others["c5"] = {"error": repr(ex)}
        line["other_configs"] = others
    if multi:
        # configs[3] and configs[2] over the ranks of this run (N = 1: a world of one, the same calls), and what RCCL saw
        for key in ("ins", "outs"):
            if hasattr(wl, key):
                delattr(wl, key)
        if not job.stub:
            job.torch.cuda.empty_cache()
        m, abandoned = guarded_multi_gpu_extras(job, args)
        line["rccl"] = m.pop("rccl", None)
        line.setdefault("other_configs", {}).update(m)
        if abandoned:
            leave_without_teardown(line)
    if not args.no_cpu_baseline and job.world == 1 and not job.stub:
        try:
            line["cpu_baseline"] = cpu_baseline(4096, 4096, wl.ss, wl.q, args.cpu_seconds)
        except Exception as ex:  # (the GPU numbers above stand on their own)
            line["cpu_baseline"] = {"error": repr(ex)}
        try:
            ref = cpu_reference_wasm(4096, 4096, wl.ss, wl.q)
        except Exception:
            ref = None
        if ref:
            line["cpu_reference"] = ref
    job.finish(line)


def run_png(job, args):
    """--workload c5: configs[4], 4096x4096 RGBA8 through the PNG row-filter stage (Adaptive strategy)
    + Adler-32 partials.  Algorithmic bytes: read 4 B/px + write (4 + 1/4096) B/px."""
    wl = PngWorkload(job)
    settled = job.settle(wl.step, args.settle_ms)
    walls, evs = job.time_blocks(wl.step, args.steps, args.warmup, args.blocks)
    if job.rank == 0 and job.world == 1:
        measure_engine_clock(job)
    if job.rank == 0 and not os.environ.get("PIXO_BENCH_ABLATION"):
        wl.check()
    if job.rank != 0:
        job.finish()
        return
    st = block_stats(walls, args.steps)
    kernel_ms = statistics.median(evs) / args.steps
    alg = wl.in_bytes + wl.out_bytes
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    traffic, src = traffic_of("c5")
    line = {"metric": "Mpixels/s PNG row filters + Adler-32 partials (Adaptive), 4096x4096 RGBA8",
            "value": round(wl.w * wl.h * job.world / (st["ms_per_step"] * 1e-3) / 1e6, 1),
            "unit": "Mpixels/s", "n_gpus": job.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": st["ms_per_step"], "ms_per_step_min": st["ms_per_step_min"], "ms_per_step_max": st["ms_per_step_max"],
            "blocks": st["blocks"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[4]: 4096x4096 RGBA8, FilterStrategy::Adaptive, rows independent", "width": wl.w, "height": wl.h,
                       "buffers_rotated": wl.nbuf, "settle_launches_before_warmup": settled,
                       "parallelism": "one process per GPU, images sharded across ranks, no collective"},
            "roofline": {"bound": bound_of(achieved / HBM_PEAK_GBPS, issue_of("c5", kernel_ms * 1e3)), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": src,
                         **issue_of("c5", kernel_ms * 1e3),
                         "kernel": "png_filter_kernel<4, true>", "algorithmic_bytes_per_launch": alg, "kernel_us_avg": round(kernel_ms * 1e3, 3)}}
    if job.world == 1 and not args.no_extras:
        with_copy(job, line["roofline"], wl.in_bytes, wl.out_bytes, kernel_ms * 1e3, line["roofline"], steps=args.steps)
    if not args.no_cpu_baseline and job.world == 1:
        import oracle_lib as O
        rows = 256  # bounded sample: 256 rows of the same image, one thread
        t1 = time.perf_counter()
        O.png_filter(wl.base[: wl.w * rows * wl.bpp], wl.w, rows, wl.bpp, O.S_ADAPTIVE)
        dt = time.perf_counter() - t1
        line["cpu_baseline"] = {"value": round(wl.w * rows / dt / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                                "sample": "first %d rows of the same 4096x4096 RGBA image, Adaptive, oracle/pixo_png_oracle.c, gcc -O2, 1 thread" % rows}
    job.finish(line)


__all__ = [n for n in dir() if not n.startswith("__")]
