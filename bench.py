#!/usr/bin/env python3
"""bench.py — Mpixels/s of the JPEG encode pixel pipeline (RGB→YCbCr→DCT→quant) on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic input that is already resident in HBM.

  --workload c2 (default, the metric; BASELINE.json configs[1]): one launch of the fused colour + DCT + quantise
      kernel over one 4096x4096 RGB8 image, q=80, 4:2:0.  Each rank rotates over enough distinct input/output
      buffers to exceed the 256 MiB Infinity Cache, so the bytes really come from and go to HBM.  N > 1: every rank
      encodes its own images (weak scaling, no data-path collective — SURVEY §8e).
  --workload c4 (configs[3]): ONE 16384x16384 image whose MCU-row bands live on the N GPUs; a step = the whole file:
      coefficient kernel + per-band entropy coding on every rank, the exchanges of pixo_amd/sharded.py (3 x i16 and a
      u64 per band over RCCL), the bodies gathered over xGMI, spliced on rank 0 (strong scaling).
  c2_444, c2_unaligned (4094 wide: rows not dword aligned), c3 (64 x 1080p, one launch), c1, c5 (PNG filters).

Timing: W warmup steps, then R blocks (--blocks) of EXACTLY K steps, each bracketed by a barrier and
torch.cuda.synchronize() on both sides, MAX over ranks per block; `ms_per_step` is the MEDIAN block (min and max
beside it).  Rank 0 prints ONE JSON line with the contract's keys plus `roofline`, `cpu_baseline`, `other_configs`.
"""
import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
C4_SHA256 = "77cc6cb69a782693c46f2024ac57ebfdfb8411148fa3cef62698c727af36c70c"  # SURVEY §8c, made by the reference

WORKLOADS = {
    #               w      h   batch ss  label
    "c2": (4096, 4096, 1, 1, "configs[1]: single 4096x4096 RGB8, q=80, 4:2:0, fused colour+DCT+quant kernel"),
    "c2_444": (4096, 4096, 1, 0, "4096x4096 RGB8, q=80, 4:4:4"),
    "c2_unaligned": (4094, 4096, 1, 1, "4094x4096 RGB8 (rows not dword aligned: funnel loads), q=80, 4:2:0"),
    "c3": (1920, 1080, 64, 1, "configs[2]: batch of 64 x 1920x1080 RGB8, q=80, 4:2:0, one launch"),
    "c1": (512, 512, 1, 1, "configs[0] shape on the GPU: 512x512 RGB8, q=80, 4:2:0"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--blocks", type=int, default=15, help="R: the K-step block is timed R times; median/min/max are reported")
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="untimed launches before the warmup steps until this much wall time has passed: after an idle "
                         "period the GPU needs ~20 ms of work to reach its steady clocks (tools/warmup_probe.py); 0 = none")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS) + ["c4", "c5"])
    ap.add_argument("--quality", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip whole_file and other_configs (A/B runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    ap.add_argument("--single-process", action="store_true",
                    help="--workload c4 only: ONE process drives the N GPUs through pixo_hip_jpeg_encode_multi (a host thread per band, every "
                         "band over its own GPU's PCIe link) instead of one rank per GPU over RCCL; pixels start in HOST memory")
    ap.add_argument("--stub", action="store_true",
                    help="plumbing test without a GPU: gloo process group, the step is a short sleep (data: 'stub')")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# process plumbing: --gpus N is honoured whichever way the script is started
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def ensure_world(args):
    """Returns (rank, local_rank, world).  `python bench.py --gpus N` with N > 1 and no launcher environment
    re-executes itself as N ranks under torch.distributed.run; a launcher whose world size differs from --gpus is an error."""
    env_world = os.environ.get("WORLD_SIZE")
    if getattr(args, "single_process", False):
        if env_world not in (None, "1"):
            raise SystemExit("bench: --single-process is ONE process for all GPUs: do not start it under a multi-rank launcher")
        return 0, 0, 1
    if env_world is None and args.gpus > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    claim_stdout()  # (not before the re-execution above: the ranks it starts inherit this process's descriptors)
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to report a number "
                         "for a different GPU count" % (args.gpus, world))
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world


class Job:
    """Process group + device + the timing protocol, shared by every workload."""

    def __init__(self, args):
        self.args = args
        self.rank, self.local_rank, self.world = ensure_world(args)
        self.stub = args.stub
        self.dist = None
        import torch
        self.torch = torch
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # TEST MODE (PIXO_BENCH_SHARE_GPU=1, tests/test_gpu_multi.py): the N ranks of a run all use GPU 0 and talk over gloo — RCCL
        # refuses two ranks on one device.  What a box with ONE GPU can check of an N-rank run: every leg's control flow and the
        # files' bytes at ranks above 0.  The line says so (`data`, `rccl.backend`); its numbers are not N-GPU numbers.
        self.share_gpu = bool(os.environ.get("PIXO_BENCH_SHARE_GPU")) and not self.stub and self.world > 1
        if self.stub:
            self.dev = self.wire = torch.device("cpu")
            self.gpu_index = None
            if self.dist is not None:
                self.dist.init_process_group(backend="gloo")
        else:
            self.gpu_index = 0 if self.share_gpu else self.local_rank
            torch.cuda.set_device(self.gpu_index)
            self.dev = torch.device("cuda", self.gpu_index)
            self.wire = torch.device("cpu") if self.share_gpu else self.dev  # where the tensors of the timing collectives live
            if self.dist is not None:
                if self.share_gpu:
                    self.dist.init_process_group(backend="gloo")
                else:
                    self.dist.init_process_group(backend="nccl", device_id=self.dev)

    def sync(self):
        if not self.stub:
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.sync()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.wire)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def settle(self, step, ms):
        """Untimed: keep the GPU busy for `ms` so that warmup and timed steps run at steady clocks."""
        n = 0
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < ms:
            for _ in range(16):
                step(n); n += 1
            self.sync()
        return n

    def time_blocks(self, step, steps, warmup, blocks, events=True):
        """W warmup steps, then `blocks` blocks of exactly `steps` steps: barrier + synchronize on both sides of every
        block, MAX over ranks.  Returns (wall seconds per block, HIP-event milliseconds per block on this rank)."""
        torch = self.torch
        for i in range(warmup):
            step(i)
        walls, evs = [], []
        n = warmup
        for _ in range(blocks):
            self.barrier()
            if events and not self.stub:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            t0 = time.perf_counter()
            for i in range(steps):
                step(n + i)
            if events and not self.stub:
                e1.record()
            self.sync()
            dt = time.perf_counter() - t0
            n += steps
            walls.append(self.max_over_ranks(dt))
            if events and not self.stub:
                evs.append(e0.elapsed_time(e1))
        self.barrier()
        return walls, evs

    def finish(self, line=None):
        """Tears the process group down, then (rank 0) prints the ONE JSON line — last, after whatever the
        runtime libraries still had in their stdio buffers (RCCL prints a version banner to stdout)."""
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
        if line is not None:
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            emit(line)


def block_stats(walls, steps):
    per = sorted(w / steps * 1e3 for w in walls)
    return {"ms_per_step": round(statistics.median(per), 5), "ms_per_step_min": round(per[0], 5), "ms_per_step_max": round(per[-1], 5),
            "blocks": len(per)}


# ------------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only)
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline(w, h, ss, quality, budget_s):
    """The oracle (C restatement, gcc -O2 -ffp-contract=off, OpenMP over MCU rows) timed on this host's cores on the same
    4096x4096 image, coefficient stage only (the work the GPU kernel does).  `value` = the median of separated single runs
    at the thread count a short probe found fastest on this box (cgroup quotas make "all logical CPUs" slower than fewer
    threads); `value_1_thread` beside it; a back-to-back burst only as a note."""
    import oracle_lib as O
    import synth
    px = synth.noise(w, h, 42)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    O.coeffs(px[: 64 * 64 * 3], 64, 64, 2, ss, quality)  # load lib
    tried = {}
    best_dt, cores = None, 1
    for th in sorted({avail, max(1, avail // 2), max(1, avail // 4), min(avail, 64), min(avail, 32), min(avail, 16)}):
        O.coeffs(px, w, h, 2, ss, quality, threads=th)  # warm the threads
        dts = []
        for _ in range(2):
            t0 = time.perf_counter()
            O.coeffs(px, w, h, 2, ss, quality, threads=th)
            dts.append(time.perf_counter() - t0)
        tried[th] = round(w * h / min(dts) / 1e6, 1)
        if best_dt is None or min(dts) < best_dt:
            best_dt, cores = min(dts), th
    # `value` = the MEDIAN of single runs at that thread count, each behind a pause: the boxes run under a cgroup CPU quota,
    # a back-to-back burst spends the quota's accumulated budget in its first repetitions and is throttled for the rest —
    # its sustained rate measured the quota, not the cores, and moved from round to round (702 -> 272 Mpixels/s for the
    # same code).  Separated runs each start with a refilled budget: comparable from run to run.  The burst stays as a note.
    pause = 0.5
    n_single = max(5, min(15, int(budget_s / (pause + best_dt))))
    singles = []
    for _ in range(n_single):
        time.sleep(pause)
        t0 = time.perf_counter()
        O.coeffs(px, w, h, 2, ss, quality, threads=cores)
        singles.append(time.perf_counter() - t0)
    singles.sort()
    reps = 8
    t_all = time.perf_counter()
    for _ in range(reps):
        O.coeffs(px, w, h, 2, ss, quality, threads=cores)
    t_all = time.perf_counter() - t_all
    out = {"value": round(w * h / statistics.median(singles) / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "kind": "port",
           "value_is": "median of %d single runs at %d threads, %.1f s apart" % (n_single, cores, pause),
           "best_single_run_Mpx_s": round(w * h / singles[0] / 1e6, 2), "worst_single_run_Mpx_s": round(w * h / singles[-1] / 1e6, 2),
           "note_burst_of_%d_back_to_back_Mpx_s" % reps: round(reps * w * h / t_all / 1e6, 2),
           "logical_cpus": avail, "probe_single_runs_Mpx_s_by_threads": tried,
           "sample": "%d x (%dx%d RGB8 noise seed 42, q=%d, %s) coefficient stage (colour+DCT+quant) "
                     "by oracle/pixo_oracle.c, gcc -O2 -ffp-contract=off, OpenMP %d threads over MCU rows"
                     % (n_single, w, h, quality, "4:2:0" if ss else "4:4:4", cores)}
    t0 = time.perf_counter()  # the reference's baseline encode_scan is single-threaded
    O.coeffs(px, w, h, 2, ss, quality, threads=1)
    out["value_1_thread"] = round(w * h / (time.perf_counter() - t0) / 1e6, 2)
    return out


def cpu_reference_wasm(w, h, ss, quality):
    """The reference's OWN code (its wasm build under node, 1 thread, whole-file encode incl.
    Huffman) on the same image, if oracle/_ref and node are available on this box."""
    wasm = os.path.join(ROOT, "oracle", "_ref", "pixo_bg.wasm")
    try:
        if not os.path.exists(wasm) or subprocess.run(["node", "--version"], capture_output=True).returncode:
            return None
        import synth
        tmp = tempfile.mkdtemp(prefix="pixo_bench_")
        inp = os.path.join(tmp, "in.bin")
        synth.noise(w, h, 42).tofile(inp)
        man = {"cases": [dict(kind="jpeg", input=inp, w=w, h=h, color_type=2, quality=quality, preset=0,
                              s420=bool(ss), repeat=4)]}
        mp = os.path.join(tmp, "m.json")
        json.dump(man, open(mp, "w"))
        r = subprocess.run(["node", "--max-old-space-size=4096", os.path.join(ROOT, "oracle", "ref_wasm.js"), mp],
                           capture_output=True, text=True, timeout=120)
        ms = json.loads(r.stdout.strip().splitlines()[0])["ms"]
        best = min(ms[1:])  # discard the JIT warm-up call
        return {"value": round(w * h / best / 1e3, 2), "unit": "Mpixels/s", "cores": 1, "kind": "reference",
                "sample": "pixo v0.4.1 wasm32 build under node (V8 JIT), whole-file encode incl. Huffman, "
                          "best of 3 warm runs on one %dx%d image" % (w, h)}
    except Exception as e:  # never let the baseline leg break the bench line
        return {"error": str(e)}


GPU_CLOCK_HZ = 2.4e9  # MI355X peak engine clock (rocminfo clockRate; MI355X_MICROARCH.md)
SIMDS = 1024          # 256 CUs x 4
CLOCK = {"hz": None}  # the engine clock under full vector load MEASURED IN THIS RUN (measure_engine_clock)


def measure_engine_clock(job):
    """pixo_hip_debug_engine_clock: shader-clock ticks over constant-clock ticks while every SIMD issues vector instructions.  The
    chip clocks down under vector load (2.0-2.4 GHz): the issue roofline's denominator is this clock, not the peak."""
    if CLOCK["hz"] is None and not job.stub:
        try:
            from pixo_amd import jpeg
            CLOCK["hz"] = jpeg.debug_engine_clock(job.torch.cuda.current_stream().cuda_stream)
        except Exception as ex:
            sys.stderr.write("bench: engine clock not measured: %r\n" % (ex,))
            CLOCK["hz"] = 0.0
    return CLOCK["hz"] or None


def _profile(kind, name):
    """A committed counter profile (profiles/<kind>_<name>.json) and whether it still describes the loaded library: the profile
    records the library version and a hash of the sources its kernel is compiled from (tools/profile_meta.py); counters of another
    build are STALE — reported as such, never as this run's."""
    path = os.path.join(ROOT, "profiles", "%s_%s.json" % (kind, name))
    d = json.load(open(path))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import profile_meta
    return d, os.path.relpath(path, ROOT), profile_meta.stale_reason(d, name)


def issue_of(name, kernel_us):
    """The VALU-ISSUE roofline of a kernel: cycles in which a SIMD's vector ALU was issuing, summed over the SIMDs — from the
    committed PMC profile profiles/issue_<name>.json (rocprofv3 --pmc SQ_ACTIVE_INST_VALU ..., tools/issue_profile.py) — over what
    1,024 SIMDs offer during the kernel time measured IN THIS RUN at the engine clock MEASURED IN THIS RUN under full vector load
    (measure_engine_clock; the 2.4 GHz peak only when that failed; `frac_issue_at_peak_clock` beside it).  Near 1: only fewer or cheaper
    vector instructions make the kernel faster, whatever its HBM fraction says.  (`valu_busy_under_counters` is the same numerator
    over the PROFILED launch's own duration, which the counters stretch: 28 us against 18 for the metric's kernel.)"""
    try:
        d, rel, stale = _profile("issue", name)
        if stale:
            return {"frac_issue": None, "counters_stale": True, "counters_stale_reason": stale, "issue_source": "profile: " + rel}
        # (older profiles: instructions x 4)
        active = d.get("active_valu_cycles_per_launch") or d["insts_valu_per_launch"] * 4.0
        clock = CLOCK["hz"] or GPU_CLOCK_HZ
        frac = active / (SIMDS * clock * kernel_us * 1e-6)
        return {"frac_issue": round(frac, 4), "frac_issue_at_peak_clock": round(active / (SIMDS * GPU_CLOCK_HZ * kernel_us * 1e-6), 4),
                "valu_insts_per_launch": d["insts_valu_per_launch"], "valu_active_cycles_per_launch": active,
                "engine_clock_GHz": round(clock / 1e9, 3),
                "engine_clock_is": "measured in this run under full vector load (pixo_hip_debug_engine_clock)" if CLOCK["hz"] else "assumed (peak)",
                "valu_busy_under_counters": d.get("valu_busy"), "counters_stale": False,
                "issue_source": "profile: " + rel + " (SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x engine clock x kernel time of this run))"}
    except Exception:
        return {}


def bound_of(frac_hbm, issue):
    """Which roofline binds: the larger of the two fractions."""
    fi = issue.get("frac_issue")
    if fi is None:
        return "hbm (issue counters missing or stale)" if issue.get("counters_stale") else "hbm"
    return "valu-issue" if fi > frac_hbm else "hbm"


def traffic_of(workload):
    """HBM bytes per launch from the committed PMC profile of this workload (separate rocprofv3 --pmc passes,
    corrected as MI355X_MICROARCH.md prescribes) — read from profiles/, NOT measured in this run; None (and the reason) when the
    profile was measured on another build of the kernel."""
    try:
        d, rel, stale = _profile("traffic", workload)
        if stale:
            return None, "STALE, not reported — %s (profile: %s)" % (stale, rel)
        return d.get("hbm_bytes_per_launch"), "profile: " + rel
    except Exception:
        return None, None


def copy_ceiling(job, in_bytes, out_bytes, steps=200, blocks=5):
    """What the memory system of THIS box, at THIS moment, gives a plain copy that reads `in_bytes` and writes `out_bytes` in the
    kernels' launch shape (pixo_hip_debug_stream_io: 192-thread workgroups, every thread R non-temporal 16-byte loads then W
    stores, R / W in {1, 2, 4, 8, 16}; the side with more bytes gets 8 per thread) — timed by the same block protocol as the
    kernel beside it.  The copy never moves fewer bytes than asked for (what it moved is reported)."""
    from pixo_amd import jpeg
    torch = job.torch
    piece = 3072
    big = max(in_bytes, out_bytes)
    wgs = max(1, -(-big // (8 * piece)))

    def pow2_at_least(x):
        v = 1
        while v < x and v < 16:
            v *= 2
        return v
    r, w = pow2_at_least(-(-in_bytes // (wgs * piece))), pow2_at_least(-(-out_bytes // (wgs * piece)))
    cin, cout = wgs * r * piece, wgs * w * piece
    nbuf = min(16, max(2, -(-(640 << 20) // (cin + cout))))  # (rotate over more than the 256 MiB Infinity Cache)
    ins = [torch.empty(cin, dtype=torch.uint8, device=job.dev).random_(0, 256) if i == 0 else torch.empty(cin, dtype=torch.uint8, device=job.dev) for i in range(nbuf)]
    for t in ins[1:]:
        t.copy_(ins[0])
    outs = [torch.empty(cout, dtype=torch.uint8, device=job.dev) for _ in range(nbuf)]
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        k = i % nbuf
        jpeg.debug_stream_io(ins[k], outs[k], wgs, r, w, stream=stream)
    _, evs = job.time_blocks(step, steps, 20, blocks)
    us = statistics.median(evs) / steps * 1e3
    del ins, outs
    torch.cuda.empty_cache()
    return {"copy_us_same_run": round(us, 3), "copy_bytes_in": cin, "copy_bytes_out": cout, "copy_shape": "%d workgroups x 192 threads, %d loads + %d stores of 16 B" % (wgs, r, w),
            "copy_GBps_same_run": round((cin + cout) / (us * 1e-6) / 1e9, 1)}


def with_copy(job, out, in_bytes, out_bytes, kernel_us, issue, steps=200):
    """Adds the same-run copy ceiling of a kernel line and lets `bound` compare like with like: the kernel's share of what a plain
    copy of its bytes gets against its share of the issue rate."""
    try:
        c = copy_ceiling(job, in_bytes, out_bytes, steps=steps)
        out.update(c)
        out["frac_of_copy_same_run"] = round(c["copy_us_same_run"] / kernel_us, 4)
        out["bound"] = bound_of(out["frac_of_copy_same_run"], issue)
        out["bound_rule"] = "larger of frac_of_copy_same_run and frac_issue"
    except BaseException as ex:  # (the kernel's number stands on its own)
        out["copy_error"] = repr(ex)
    return out


# ------------------------------------------------------------------------------------------------------------------
# the coefficient kernel (c2, c2_444, c2_unaligned, c3, c1)
# ------------------------------------------------------------------------------------------------------------------
class CoeffWorkload:
    def __init__(self, job, name, quality):
        import numpy as np
        import synth
        from pixo_amd import jpeg
        self.job, self.name, self.q, self.jpeg, self.np = job, name, quality, jpeg, np
        torch = job.torch
        self.w, self.h, self.batch, self.ss, self.label = WORKLOADS[name]
        w, h, batch, ss = self.w, self.h, self.batch, self.ss
        self.yb, self.cbn = jpeg.coefficient_geometry(w, h, 2, ss)
        self.in_bytes = w * h * 3 * batch
        self.out_bytes = (self.yb + 2 * self.cbn) * 128 * batch
        # rotate over enough buffer sets that the working set exceeds the 256 MiB Infinity Cache
        self.nbuf = min(64, max(2, -(-(640 << 20) // (self.in_bytes + self.out_bytes))))
        self.base = synth.noise(w, h, 42 + job.rank)
        if job.stub:
            self.ins = self.outs = None
            return
        host = torch.from_numpy(np.ascontiguousarray(self.base))
        dev = job.dev
        self.ins, self.outs = [], []
        for i in range(self.nbuf):
            t = host.to(dev)
            if batch > 1:
                t = t.repeat(batch)
            t = t ^ torch.tensor(i & 0xFF, dtype=torch.uint8, device=dev) if i else t  # distinct content per buffer
            self.ins.append(t.contiguous())
            self.outs.append((torch.empty((batch * self.yb, 64), dtype=torch.int16, device=dev),
                              torch.empty((batch * self.cbn, 64), dtype=torch.int16, device=dev),
                              torch.empty((batch * self.cbn, 64), dtype=torch.int16, device=dev)))
        self.stream = torch.cuda.current_stream().cuda_stream

    def step(self, i):
        if self.job.stub:
            time.sleep(2e-5)
            return
        k = i % self.nbuf
        y, cb, cr = self.outs[k]
        self.jpeg.coefficients_device(self.ins[k], self.w, self.h, 2, self.ss, self.q, y, cb, cr, batch=self.batch, stream=self.stream)

    def check(self):
        """correctness inside the bench: buffer 0 against the oracle on a 64-row strip (Y, Cb and Cr)"""
        import oracle_lib as O
        np = self.np
        strip_h = 64
        oy, ocb, ocr = O.coeffs(self.base[: self.w * strip_h * 3], self.w, strip_h, 2, self.ss, self.q)
        self.step(0)
        self.job.sync()
        gy = self.outs[0][0][: oy.shape[0]].cpu().numpy()
        gcb = self.outs[0][1][: ocb.shape[0]].cpu().numpy()
        gcr = self.outs[0][2][: ocr.shape[0]].cpu().numpy()
        if not (np.array_equal(gy, oy) and np.array_equal(gcb, ocb) and np.array_equal(gcr, ocr)):
            raise SystemExit("bench: GPU coefficients differ from the oracle — refusing to report a number")

    def roofline(self, kernel_ms, copy_ms=None):
        alg = self.in_bytes + self.out_bytes  # 3 B/px read + 3 B/px written (4:2:0); 3 + 6 for 4:4:4
        achieved = alg / (kernel_ms * 1e-3) / 1e9
        traffic, src = traffic_of(self.name)
        issue = issue_of(self.name, kernel_ms * 1e3)
        # `bound` is COMPUTED: the larger of the two fractions of this run (HBM bytes against 8 TB/s, vector instructions
        # against what 1,024 SIMDs issue) names the roofline that binds
        r = {"bound": bound_of(achieved / HBM_PEAK_GBPS, issue), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
             "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": src, **issue,
             "kernel": "jpeg_coeffs_kernel<%s, %s>" % ("M420" if self.ss else "M444", "L_FUNNEL" if self.w * 3 % 4 else "L_ALIGNED"),
             "algorithmic_bytes_per_launch": alg, "kernel_us_avg": round(kernel_ms * 1e3, 3),
             "read_only_frac_of_peak": round(self.in_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        if copy_ms:
            # the plain copy of the same bytes in the same launch shape, timed in THIS run by the same block protocol
            # (pixo_hip_debug_stream_copy): what the memory system of this box, at this moment, gives 50 MB in + 50 MB out
            r["copy_us_same_run"] = round(copy_ms * 1e3, 3)
            r["copy_GBps_same_run"] = round(alg / (copy_ms * 1e-3) / 1e9, 1)
            r["copy_frac_of_peak_same_run"] = round(alg / (copy_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
            r["kernel_over_copy_same_run"] = round(kernel_ms / copy_ms, 4)
            r["frac_of_copy_same_run"] = round(copy_ms / kernel_ms, 4)
            # with the memory system's own ceiling measured in the same run, `bound` compares like with like: the kernel's share of
            # what a plain copy of its bytes gets against its share of the issue rate (a kernel at 0.99 of the copy and 0.82 of the
            # issue rate is bound by the memory system, although 0.71 of the 8 TB/s PEAK is the smaller number)
            r["bound"] = bound_of(copy_ms / kernel_ms, issue)
            r["bound_rule"] = "larger of frac_of_copy_same_run and frac_issue"
        return r

    def copy_step_factory(self):
        """A plain copy of this workload's bytes (4:2:0: as many out as in) over the same rotating input buffers."""
        if self.in_bytes != self.out_bytes or self.in_bytes % 24576:
            return None
        torch = self.job.torch
        outs = [torch.empty(self.in_bytes, dtype=torch.uint8, device=self.job.dev) for _ in range(self.nbuf)]
        jpeg, ins, n, nb, stream = self.jpeg, self.ins, self.in_bytes, self.nbuf, self.stream

        def step(i):
            k = i % nb
            jpeg.debug_stream_copy(ins[k], outs[k], n, stream=stream)
        step.outs = outs
        return step


QUICK_SETTLE_MS = 60.0  # the extras run behind host-bound phases (allocations, uploads, the oracle check): the clocks have dropped


def quick_kernel(job, name, q, steps=200, blocks=7):
    """One of the other configurations, measured the way the metric is: the workload's own launches keep the GPU busy for
    QUICK_SETTLE_MS first (each of these follows a host-bound phase — building the buffers, the oracle check — during which
    the clocks fall), then warmup, then the median of `blocks` blocks of `steps` steps; kernel time from HIP events."""
    wl = CoeffWorkload(job, name, q)
    wl.check()
    job.settle(wl.step, QUICK_SETTLE_MS)
    _, evs = job.time_blocks(wl.step, steps, 20, blocks)
    kernel_ms = statistics.median(evs) / steps
    r = wl.roofline(kernel_ms)
    out = {"workload": wl.label, "kernel_us": r["kernel_us_avg"], "Mpixels_per_s": round(wl.w * wl.h * wl.batch / kernel_ms / 1e3, 1),
           "achieved_GBps": r["achieved"], "frac": r["frac"], "bound": r["bound"], "steps": steps, "blocks": blocks, "settle_ms": QUICK_SETTLE_MS}
    for key in ("frac_issue", "valu_insts_per_launch", "issue_source", "counters_stale", "counters_stale_reason", "engine_clock_GHz", "valu_busy_under_counters",
                "traffic", "traffic_source"):
        if key in r:
            out[key] = r[key]
    in_bytes, out_bytes = wl.in_bytes, wl.out_bytes
    del wl
    job.torch.cuda.empty_cache()
    # the plain copy of the same bytes (50 -> 100 MB for 4:4:4, 398 -> 401 MB for the batch ...) right behind the kernel's blocks
    with_copy(job, out, in_bytes, out_bytes, out["kernel_us"], r, steps=max(20, min(steps, int(2e5 / max(out["kernel_us"], 1.0)))))
    return out


def quick_png(job, steps=100, blocks=7):
    wl = PngWorkload(job)
    wl.check()
    job.settle(wl.step, QUICK_SETTLE_MS)
    _, evs = job.time_blocks(wl.step, steps, 20, blocks)
    kernel_ms = statistics.median(evs) / steps
    alg = wl.in_bytes + wl.out_bytes
    issue = issue_of("c5", kernel_ms * 1e3)
    frac = alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
    out = {"workload": "configs[4]: 4096x4096 RGBA8 PNG row filters (Adaptive) + Adler-32 partials", "kernel_us": round(kernel_ms * 1e3, 3),
           "Mpixels_per_s": round(4096 * 4096 / kernel_ms / 1e3, 1), "achieved_GBps": round(alg / (kernel_ms * 1e-3) / 1e9, 1),
           "frac": round(frac, 4), "bound": bound_of(frac, issue), **issue, "steps": steps, "blocks": blocks}
    traffic, src = traffic_of("c5")
    out["traffic"], out["traffic_source"] = traffic, src
    in_bytes, out_bytes = wl.in_bytes, wl.out_bytes
    del wl
    job.torch.cuda.empty_cache()
    with_copy(job, out, in_bytes, out_bytes, out["kernel_us"], issue, steps=steps)
    return out


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


def config_1(job, q):
    """configs[0]: a single 512x512 RGB8 image, q=80, 4:2:0 — the reference's own CPU-runnable case ("plumbing, no GPU").  Three
    numbers side by side: the CPU port of the whole encode (oracle/pixo_oracle.c, one thread, median of 9 files; the checker, used
    here as the CPU leg only), the GPU library on the same pixels (host pixels -> file bytes, median of 100 calls; compared with
    the CPU's bytes and with the reference-made golden of SURVEY §8c), and the coefficient kernel alone on that shape."""
    import numpy as np
    import oracle_lib as O
    import synth
    from pixo_amd import jpeg
    w = h = 512
    px = np.ascontiguousarray(synth.noise(w, h, 42)).reshape(-1)
    oo = O.make_options(w, h, 2, q, 1)
    want = O.encode(px, oo)
    tc = []
    for _ in range(9):
        t1 = time.perf_counter()
        O.encode(px, oo)
        tc.append(time.perf_counter() - t1)
    opts = jpeg.JpegOptions.builder(w, h).quality(q).subsampling(jpeg.Subsampling.S420).build()
    got = jpeg.encode(px, opts)
    if got != want:
        raise SystemExit("bench: the 512x512 file differs from the oracle's — refusing to report a number")
    golden = hashlib.sha256(got).hexdigest() == "128275e652c0e640e9bde2360d7a39c58f951f998b00319c07a6209cbe6dd159" if q == 80 else None
    for _ in range(10):
        jpeg.encode(px, opts)
    tg = []
    for _ in range(100):
        t1 = time.perf_counter()
        jpeg.encode(px, opts)
        tg.append(time.perf_counter() - t1)
    k = quick_kernel(job, "c1", q, steps=200, blocks=5)
    cpu_ms, gpu_us = sorted(tc)[4] * 1e3, sorted(tg)[50] * 1e6
    return {"workload": "configs[0]: single 512x512 RGB8 -> JPEG q=%d 4:2:0 (noise, seed 42)" % q, "file_bytes": len(got),
            "file_equals_reference_golden_sha256": golden,
            "cpu_whole_file_ms": round(cpu_ms, 3), "cpu_Mpixels_per_s": round(w * h / cpu_ms / 1e3, 2), "cpu_is": "oracle/pixo_oracle.c (C port of the reference's encode), 1 thread",
            "gpu_whole_file_us_host_pixels_to_bytes": round(gpu_us, 1), "gpu_Mpixels_per_s_whole_file": round(w * h / gpu_us, 1),
            "coefficient_kernel": {key: k[key] for key in ("kernel_us", "frac", "Mpixels_per_s", "copy_us_same_run", "frac_of_copy_same_run") if key in k}}


def batch_whole_files(job, q, n_batches=7):
    """configs[2] as WHOLE FILES: 64 x 1920x1080 device-resident images -> 64 JPEG files back to back in the caller's pinned
    arena (pixo_hip_jpeg_encode_batch_device_into): one coefficient launch, the images as segments of the two single-pass
    entropy kernels, every file copied from the device straight to its final place."""
    import numpy as np
    import synth
    from pixo_amd import jpeg
    torch = job.torch
    w, h, n = 1920, 1080, 64
    base = torch.from_numpy(np.ascontiguousarray(synth.noise(w, h, 42))).to(job.dev)
    d = torch.cat([base ^ torch.tensor(i, dtype=torch.uint8, device=job.dev) for i in range(n)]).contiguous()
    opts = jpeg.JpegOptions.builder(w, h).quality(q).subsampling(jpeg.Subsampling.S420).build()
    arena = torch.empty(n * w * h, dtype=torch.uint8).pin_memory()
    offs, lens = jpeg.encode_batch_device_into(arena, d, opts, n)
    import oracle_lib as O
    first = arena[offs[0]: offs[0] + lens[0]].numpy().tobytes()
    if first != O.encode(synth.noise(w, h, 42), O.make_options(w, h, 2, q, 1)):
        raise SystemExit("bench: batch file 0 differs from the oracle's — refusing to report a number")
    ts = []
    for _ in range(n_batches):
        t1 = time.perf_counter()
        offs, lens = jpeg.encode_batch_device_into(arena, d, opts, n)
        ts.append(time.perf_counter() - t1)
    dt = sorted(ts)[len(ts) // 2]
    # the drop-in shape (a pixo caller's `encode()` per image returns a Vec it owns): `pixo_hip_jpeg_encode_batch_device` hands out 64
    # blocks the caller owns until pixo_hip_free — timed as a C caller sees it (call + the 64 frees); file 0 checked
    import ctypes
    tb = []
    for rep in range(6):
        t1 = time.perf_counter()
        fp, fl = jpeg.encode_batch_device_raw(d, opts, n)
        t2 = time.perf_counter()
        if rep == 0 and ctypes.string_at(fp[0], fl[0]) != first:
            raise SystemExit("bench: malloc'd batch file 0 differs from the oracle's — refusing to report a number")
        t3 = time.perf_counter()
        jpeg.free_files(fp, n)
        tb.append((t2 - t1) + (time.perf_counter() - t3))
    tb = tb[1:]  # (the first call allocates the blocks; every later one gets them back from pixo_hip_free)
    # the same batch with photograph-like content (synth.photo, ~1.3 bit/px: the users' case; every image its own copy in HBM)
    photo = {}
    try:
        dp = torch.from_numpy(np.ascontiguousarray(synth.photo(w, h, 42))).to(job.dev).repeat(n).contiguous()
        for _ in range(2):
            offs_p, lens_p = jpeg.encode_batch_device_into(arena, dp, opts, n)
        tp = []
        for _ in range(n_batches):
            t1 = time.perf_counter()
            offs_p, lens_p = jpeg.encode_batch_device_into(arena, dp, opts, n)
            tp.append(time.perf_counter() - t1)
        photo = {"ms_per_batch_photo": round(sorted(tp)[len(tp) // 2] * 1e3, 3), "file_bytes_total_photo": int(sum(lens_p))}
        del dp
    except Exception as ex:
        photo = {"photo_error": repr(ex)}
    # DEVICE time of the batch (the PCIe-bound 1.7 ms hides the kernels): the product's kernels for the 64 images enqueued back to back
    # (pixo_hip_debug_scan_device_async_batch: no waits, nothing delivered), HIP events on the launch stream — the default form
    # (coefficient kernel + scan_code + stuffing kernel over the batch) and the fused pixel -> scan kernel with every image a segment
    # (debug switch fused_batch; slower on a launch of several generations, which is why it is not the default)
    device = {}
    try:
        stream = torch.cuda.current_stream().cuda_stream
        for name, sw in (("default_two_kernel_form", None), ("fused_kernel_every_image_a_segment", "fused_batch")):
            jpeg.debug_configure(sw)
            form = jpeg.debug_scan_device_async(d, opts, stream=stream, batch=n)
            job.sync()
            _, evs = job.time_blocks(lambda i: jpeg.debug_scan_device_async(d, opts, stream=stream, batch=n), 10, 4, 5)
            us = statistics.median(evs) / 10 * 1e3
            device[name] = {"device_us_per_batch": round(us, 1), "fused": bool(form),
                            "frac_hbm_pixels_plus_files": round((n * w * h * 3 + int(sum(lens))) / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)}
        jpeg.debug_configure(None)
    except Exception as ex:
        jpeg.debug_configure(None)
        device = {"error": repr(ex)}
    del d, arena
    torch.cuda.empty_cache()
    return {**photo, "device_time": device, "workload": "configs[2] whole files: 64 x 1920x1080 RGB8 noise, q=%d, 4:2:0 -> 64 files in one pinned arena" % q,
            "ms_per_batch": round(dt * 1e3, 3), "ms_per_batch_min": round(min(ts) * 1e3, 3), "Mpixels_per_s": round(w * h * n / dt / 1e6, 1),
            "file_bytes_total": int(sum(lens)), "ms_per_batch_as_64_malloced_files": round(sorted(tb)[len(tb) // 2] * 1e3, 3),
            "malloced_files_are": "64 blocks from the library's pinned pool, owned by the caller until pixo_hip_free; call + frees timed, steady state (median of 5 after the first)",
            "path": "pixo_hip_jpeg_encode_batch_device_into"}


def small_files(wl):
    """Not `value`: the latency of ONE small image, host pixels -> file bytes (pixo_hip_jpeg_encode_jpeg, the wasm entry's shape:
    src/wasm.rs:113-142), per preset — 0 baseline (one kernel + one wait), 1 optimised tables, 2 trellis (eight lanes per block
    at these sizes) + progressive + optimised tables.  Median of 100 calls each; the first file of every kind against the oracle."""
    out = {}
    try:
        import numpy as np
        import oracle_lib as O
        import synth
        jpeg = wl.jpeg
        for (w, h) in ((64, 64), (512, 512)):
            px = np.ascontiguousarray(synth.noise(w, h, 42)).reshape(-1)
            row = {}
            for preset in (0, 1, 2):
                fn = lambda: jpeg.encode_jpeg(px, w, h, 2, wl.q, preset, True)
                first = bytes(fn())
                if first != bytes(O.encode_flat(px, w, h, 2, wl.q, preset, True)):
                    raise SystemExit("bench: a small preset-%d file differs from the oracle's — refusing to report a number" % preset)
                for _ in range(10):
                    fn()
                ts = []
                for _ in range(100):
                    t1 = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t1)
                row["preset%d_us" % preset] = round(sorted(ts)[50] * 1e6, 1)
                row["preset%d_bytes" % preset] = len(first)
            out["%dx%d" % (w, h)] = row
        out["what"] = "one image, host pixels -> bytes (pixo_hip_jpeg_encode_jpeg), noise, q=%d 4:2:0, median of 100 calls" % wl.q
    except SystemExit:
        raise
    except Exception as ex:  # the metric line must not depend on this extra
        out = {"error": repr(ex)}
    return out


def whole_file(job, wl):
    """Not `value`: the whole file (the fused pixel -> scan kernel + copy of the file to the host) from
    device-resident pixels, reported beside the kernel-only metric."""
    torch, jpeg = job.torch, wl.jpeg
    opts = jpeg.JpegOptions.builder(wl.w, wl.h).quality(wl.q).subsampling(jpeg.Subsampling(wl.ss)).build()
    try:
        pinned = torch.empty(wl.in_bytes // 2 + (1 << 16), dtype=torch.uint8).pin_memory()  # (64 B per block + 10 KB or more: the library may write it piece by piece)
        nbytes = jpeg.encode_device_into(pinned, wl.ins[0], opts)
        n_files, ts, tb = 15, [], []
        for i in range(n_files):
            t1 = time.perf_counter()
            nbytes = jpeg.encode_device_into(pinned, wl.ins[i % wl.nbuf], opts)
            ts.append(time.perf_counter() - t1)
        for i in range(7):
            t1 = time.perf_counter()
            jpeg.encode_device(wl.ins[i % wl.nbuf], opts)
            tb.append(time.perf_counter() - t1)
        dt, dtb = sorted(ts)[n_files // 2], sorted(tb)[3]
        host_px = torch.from_numpy(wl.base.copy())  # pageable host pixels, as pixo::jpeg::encode's caller has them
        th = []
        for i in range(7):
            t1 = time.perf_counter()
            jpeg.encode_into_buffer(pinned.numpy(), host_px.numpy(), opts)
            th.append(time.perf_counter() - t1)
        # the same for SMOOTH content (benches/comparison.rs:32 `generate_gradient_image`, SURVEY §8d's secondary input): the file is
        # 0.3 MB instead of 11 MB, so this is the kernels' and the call's latency, not PCIe
        smooth = {}
        try:
            import synth
            d_g = torch.from_numpy(synth.gradient_rgb(wl.w, wl.h)).to(job.dev)
            for _ in range(3):
                nb_g = jpeg.encode_device_into(pinned, d_g, opts)
            tg = []
            for _ in range(15):
                t1 = time.perf_counter()
                nb_g = jpeg.encode_device_into(pinned, d_g, opts)
                tg.append(time.perf_counter() - t1)
            for _ in range(2):  # (the context predicts the next file's size from the last one: back to the metric's content)
                jpeg.encode_device_into(pinned, wl.ins[0], opts)
            smooth = {"ms_per_image_gradient": round(sorted(tg)[7] * 1e3, 3), "file_bytes_gradient": int(nb_g)}
            # ... and for PHOTOGRAPH-LIKE content (synth.photo: structure at several scales + a little sensor noise, ~1.3 bit/px
            # at q = 80 — what users encode; noise and the gradient only bracket it)
            d_p = torch.from_numpy(synth.photo(wl.w, wl.h, 42)).to(job.dev)
            for _ in range(3):
                nb_p = jpeg.encode_device_into(pinned, d_p, opts)
            tp = []
            for _ in range(15):
                t1 = time.perf_counter()
                nb_p = jpeg.encode_device_into(pinned, d_p, opts)
                tp.append(time.perf_counter() - t1)
            smooth["ms_per_image_photo"] = round(sorted(tp)[7] * 1e3, 3)
            smooth["file_bytes_photo"] = int(nb_p)
            smooth["bits_per_pixel_photo"] = round(nb_p * 8 / (wl.w * wl.h), 3)
            # the DEVICE time per file (pixo_hip_debug_scan_device_async: the product's kernel for one baseline file — pixels -> the
            # finished, stuffed scan in ONE kernel — enqueued back to back, HIP events on the launch stream, no waits,
            # no PCIe): K files between two events, median of the blocks.  frac = (pixels read + file written) / time / 8 TB/s.
            dev = {}
            for name, d_img, nb in (("noise", wl.ins[0], nbytes), ("photo", d_p, nb_p), ("gradient", d_g, nb_g)):
                form = jpeg.debug_scan_device_async(d_img, opts, stream=wl.stream)
                job.sync()
                _, evs = job.time_blocks(lambda i, d_img=d_img: jpeg.debug_scan_device_async(d_img, opts, stream=wl.stream), 50, 10, 5)
                us = statistics.median(evs) / 50 * 1e3
                dev[name] = {"device_us_per_file": round(us, 2), "frac_hbm_pixels_plus_file": round((wl.in_bytes + nb) / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                             "kernels": "pixels_code_kernel (one kernel: pixels -> stuffed scan)" if form else "jpeg_coeffs + scan_code + stuff_fused"}
                issue = issue_of("pixels_code_" + name, us) if form else {}
                dev[name].update({k: v for k, v in issue.items() if k in ("frac_issue", "valu_insts_per_launch", "counters_stale", "counters_stale_reason", "engine_clock_GHz")})
                tr, tr_src = traffic_of("pixels_code_" + name)
                dev[name]["traffic"], dev[name]["traffic_source"] = tr, tr_src
                with_copy(job, dev[name], wl.in_bytes, int(nb), us, issue, steps=100)
            smooth["device_time"] = dev
            # the other presets' files: progressive scans (prog_code_kernel: one load and one walk of a block for all
            # scans of its component) and preset 2 (trellis + progressive + optimised tables), same pixels, same pinned buffer
            b = lambda: jpeg.JpegOptions.builder(wl.w, wl.h).quality(wl.q).subsampling(jpeg.Subsampling(wl.ss))
            prog = {}
            for name, o2, d_img in (("progressive_noise", b().progressive(True).build(), wl.ins[0]),
                                    ("progressive_photo", b().progressive(True).build(), d_p),
                                    ("progressive_gradient", b().progressive(True).build(), d_g),
                                    ("preset2_noise", b().progressive(True).trellis_quant(True).optimize_huffman(True).build(), wl.ins[0]),
                                    ("preset2_photo", b().progressive(True).trellis_quant(True).optimize_huffman(True).build(), d_p)):
                for _ in range(3):
                    nb2 = jpeg.encode_device_into(pinned, d_img, o2)
                t2 = []
                for _ in range(11):
                    t1 = time.perf_counter()
                    nb2 = jpeg.encode_device_into(pinned, d_img, o2)
                    t2.append(time.perf_counter() - t1)
                prog["ms_per_image_" + name] = round(sorted(t2)[5] * 1e3, 3)
                prog["file_bytes_" + name] = int(nb2)
            smooth["other_presets"] = prog
            for _ in range(2):
                jpeg.encode_device_into(pinned, wl.ins[0], opts)
            del d_g, d_p
        except Exception as ex:
            smooth = {"gradient_error": repr(ex)}
        return {"value": round(wl.w * wl.h / dt / 1e6, 1), "unit": "Mpixels/s", "ms_per_image": round(dt * 1e3, 3), **smooth,
                "whole_file_from_host_ms": round(sorted(th)[3] * 1e3, 3), "whole_file_from_host_min_ms": round(min(th) * 1e3, 3),
                "ms_per_image_min": round(min(ts) * 1e3, 3), "file_bytes": int(nbytes), "ms_per_image_as_python_bytes": round(dtb * 1e3, 3),
                "path": "device-resident pixels -> ONE kernel: colour, DCT, quantiser, Huffman walk, bit placement, 0xFF stuffing (no coefficient tuple, no "
                        "packed stream in HBM) -> file in the caller's pinned host buffer (pixo_hip_jpeg_encode_device_into)"}
    except Exception as ex:  # the metric line must not depend on this extra
        return {"error": repr(ex)}


# ------------------------------------------------------------------------------------------------------------------
# c5: PNG row filters
# ------------------------------------------------------------------------------------------------------------------
class PngWorkload:
    def __init__(self, job):
        import synth
        from pixo_amd import png
        torch = job.torch
        self.job, self.png = job, png
        self.w = self.h = 4096
        self.bpp = 4
        self.base = synth.rgba_noise_alpha1(self.w, self.h, 42 + job.rank)
        self.in_bytes, self.out_bytes = self.w * self.h * self.bpp, png.filtered_size(self.w, self.h, self.bpp)
        self.nbuf = 5  # 5 x 134 MB > Infinity Cache
        host = torch.from_numpy(self.base)
        dev = job.dev
        self.ins = [(host.to(dev) ^ torch.tensor(i, dtype=torch.uint8, device=dev)).contiguous() for i in range(self.nbuf)]
        self.outs = [torch.empty(self.out_bytes, dtype=torch.uint8, device=dev) for _ in range(self.nbuf)]
        self.sums = [torch.zeros(2 * self.h, dtype=torch.int64, device=dev) for _ in range(self.nbuf)]
        self.scratch = torch.zeros(4, dtype=torch.int32, device=dev)
        self.stream = torch.cuda.current_stream().cuda_stream

    def step(self, i):
        k = i % self.nbuf
        self.png.apply_filters_async(self.ins[k], self.w, self.h, self.bpp, self.outs[k], self.sums[k], self.scratch,
                                     self.png.FilterStrategy.ADAPTIVE, 0, self.stream)

    def check(self):
        """buffer 0 against the reference-made vector of SURVEY §8c (rank 0's input is that very image)"""
        import numpy as np
        self.step(0)
        self.job.sync()
        adler = self.png.adler32_from_row_sums(self.sums[0].cpu().numpy().view(np.uint64), self.w, self.h, self.bpp)
        digest = hashlib.sha256(self.outs[0].cpu().numpy().tobytes()).hexdigest()
        if self.job.rank == 0 and (adler != 0x90CC12E3 or not digest.startswith("240e005d4da54561")):
            raise SystemExit("bench: filtered stream differs from the reference's — refusing to report a number")


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


C3_IMAGE0_SHA256 = "d1811ba1761f6b2a76d7f2c3d43418784f38909e0b20af631d5ead73e7d9436a"  # noise(1920,1080,42), made by the reference


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


MULTI_LEGS_DEADLINE_S = 240.0  # all multi-GPU legs together (they take ~3 s on one GPU); the metric line must not wait longer


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


_LINE_OUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner when its first
    communicator is made): from here on file descriptor 1 IS stderr, and only `emit` holds the real stdout."""
    global _LINE_OUT
    if _LINE_OUT is None:
        sys.stdout.flush()
        _LINE_OUT = os.fdopen(os.dup(1), "w", encoding="utf-8")
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    out = _LINE_OUT if _LINE_OUT is not None else sys.stdout
    out.write(json.dumps(line, ensure_ascii=False) + "\n")
    out.flush()


def main():
    args = parse()
    job = Job(args)
    if args.workload == "c4" and args.single_process:
        return run_c4_single_process(job, args)
    if args.workload == "c5":
        return run_png(job, args)
    if args.workload == "c4":
        return run_c4(job, args)
    return run_coeffs(job, args)


if __name__ == "__main__":
    main()
