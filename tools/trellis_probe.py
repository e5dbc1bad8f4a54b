"""N preset-2-style encodes (progressive + trellis) of ONE device-resident 4096x4096 image back to back: the workload
behind `trellis_kernel`'s per-launch duration list (117 -> 344 us on identical input) and its PMC passes.
  python tools/trellis_probe.py [n] [kind]            the loop (run it under rocprofv3 --kernel-trace / --pmc)
  python tools/trellis_probe.py --list <kernel_trace.csv> [substring]   per-dispatch durations of one kernel, in launch order
"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))

if len(sys.argv) > 1 and sys.argv[1] == "--list":
    import csv
    pat = sys.argv[3] if len(sys.argv) > 3 else "trellis_kernel"
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if pat in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    print("%d dispatches of %s: min %.1f  median %.1f  mean %.1f  max %.1f us" % (len(d), pat, min(d), sorted(d)[len(d) // 2], sum(d) / len(d), max(d)))
    for i, r in enumerate(rows):
        gap = (int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3 if i else 0.0
        print("  #%03d  start %10.1f us   duration %7.1f us   idle since the previous one ended %9.1f us" % (i, (int(r["Start_Timestamp"]) - t0) / 1e3, d[i], gap))
    sys.exit(0)

import numpy as np, torch
import synth
from pixo_amd import jpeg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
kind = sys.argv[2] if len(sys.argv) > 2 else "noise"
w = h = 4096
px = synth.noise(w, h, 42) if kind == "noise" else synth.gradient_rgb(w, h)
d = torch.from_numpy(px).to("cuda:0"); torch.cuda.synchronize()
o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420).progressive(True).trellis_quant(True).build()
pinned = torch.empty(w * h * 3 // 2, dtype=torch.uint8).pin_memory()
jpeg.encode_device_into(pinned, d, o)
ts = []
for _ in range(n):
    t0 = time.perf_counter()
    jpeg.encode_device_into(pinned, d, o)
    ts.append(time.perf_counter() - t0)
ts.sort()
print("progressive+trellis 4096x4096 %s: median %.3f ms, min %.3f, max %.3f over %d files" % (kind, ts[n // 2] * 1e3, ts[0] * 1e3, ts[-1] * 1e3, n))
