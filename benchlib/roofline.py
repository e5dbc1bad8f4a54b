"""bench.py's rooflines: counters from profiles/ (with the staleness guard), the engine clock measured in the run, the same-run copy ceiling."""
from .common import *  # noqa: F401,F403


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


__all__ = [n for n in dir() if not n.startswith("__")]
