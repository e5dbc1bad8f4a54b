"""Soak: T threads encode randomised cases (tests/fresh_cases.py: JPEG presets 0-2 and PNG filters; every 7th case a larger
image) for SECONDS, every result compared with the oracle; host RSS and free device memory are sampled every few seconds —
a leak shows as a trend, not a plateau.  Threads end and new ones start every `generation` cases (contexts are parked and
re-used), and the run ends with pixo_hip_trim.     python tools/soak.py [seconds] [threads]"""
import os, sys, threading, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import fresh_cases as F
import oracle_lib as O
from pixo_amd import jpeg, png
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
O.lib()
stop_at = time.time() + seconds
lock = threading.Lock()
stats = dict(cases=0, bad=0, pixels=0)
PNG = {0: (png.FilterStrategy.ADAPTIVE_FAST, png.NO_RAYON, O.S_ADAPTIVE_FAST, True), 1: (png.FilterStrategy.ADAPTIVE, 0, O.S_ADAPTIVE, False),
       2: (png.FilterStrategy.BIGRAMS, 0, O.S_BIGRAMS, False)}


def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024.0
    return 0.0


def one(i):
    if i % 2:
        o, px = F.png_case_of(i, max_side=200 if i % 7 else 900)
        s, fl, os_, st = PNG[o["preset"]]
        bpp = F.PNG_BPP[o["color_type"]]
        got, gad = png.apply_filters(px, o["w"], o["h"], bpp, s, fl)
        want, wad = O.png_filter(px, o["w"], o["h"], bpp, os_, st)
        ok = np.array_equal(got, want) and gad == wad
    else:
        o, px = F.case_of(i, max_side=320 if i % 7 else 1500)
        a = bytes(jpeg.encode_jpeg(px, o["w"], o["h"], o["color_type"], o["quality"], o["preset"], o["s420"]))
        b = bytes(O.encode_flat(px, o["w"], o["h"], o["color_type"], o["quality"], o["preset"], o["s420"]))
        ok = a == b
    with lock:
        stats["cases"] += 1; stats["pixels"] += o["w"] * o["h"]
        if not ok:
            stats["bad"] += 1; print("MISMATCH", o, flush=True)


def worker(first, step, generation=150):
    i = first
    for _ in range(generation):
        if time.time() >= stop_at: return
        one(i); i += step


def spawner(t):
    gen = 0
    while time.time() < stop_at:
        th = threading.Thread(target=worker, args=(1_000_000 + t + gen * 150 * threads, threads)); th.start(); th.join(); gen += 1


torch.cuda.init()
free0, total = torch.cuda.mem_get_info()
print("start: rss %.0f MB, device free %.0f MB" % (rss_mb(), free0 / 2**20), flush=True)
ts = [threading.Thread(target=spawner, args=(t,)) for t in range(threads)]
for t in ts: t.start()
t0 = time.time(); samples = []
while any(t.is_alive() for t in ts):
    time.sleep(min(10.0, max(1.0, seconds / 12)))
    free, _ = torch.cuda.mem_get_info()
    samples.append((time.time() - t0, rss_mb(), free / 2**20, stats["cases"]))
    print("t %5.0f s: cases %7d, rss %.0f MB, device free %.0f MB" % samples[-1][:1] + samples[-1][3:] + samples[-1][1:3] if False else
          "t %5.0f s: cases %7d, rss %.0f MB, device free %.0f MB" % (samples[-1][0], samples[-1][3], samples[-1][1], samples[-1][2]), flush=True)
for t in ts: t.join()
jpeg.trim()
free1, _ = torch.cuda.mem_get_info()
print("end: %d cases (%.0f Mpixels) on %d threads in %.0f s, %d mismatches, single-pass fallbacks %d; after trim: rss %.0f MB, device free %.0f MB (start %.0f)" %
      (stats["cases"], stats["pixels"] / 1e6, threads, time.time() - t0, stats["bad"], jpeg.lookback_fallbacks(), rss_mb(), free1 / 2**20, free0 / 2**20))
half = [s for s in samples if s[0] > seconds / 2]
if len(half) >= 2:
    print("second half of the run: rss %+.0f MB, device free %+.0f MB" % (half[-1][1] - half[0][1], half[-1][2] - half[0][2]))
sys.exit(1 if stats["bad"] else 0)
