"""bench.py's CPU legs (rank 0, N = 1 only): the oracle's C port of the coefficient stage on the host cores (`cpu_baseline`), the reference's own wasm
build where node is present (`cpu_reference`).  The oracle is the checker and the CPU baseline — never the thing measured as the product."""
from .common import *  # noqa: F401,F403


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


__all__ = [n for n in dir() if not n.startswith("__")]
