"""One-off: latency of one small image per wasm preset (0 baseline, 1 optimised tables, 2 trellis + progressive + optimised),
host pixels -> bytes, median of 200 calls.   python tools/small_latency_presets.py"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import synth
from pixo_amd import jpeg
for (w, h) in ((64, 64), (512, 512), (1024, 1024)):
    for kind in ("noise", "gradient"):
        px = synth.noise(w, h, 42) if kind == "noise" else synth.gradient_rgb(w, h)
        row = []
        for preset in (0, 1, 2):
            fn = lambda: jpeg.encode_jpeg(px, w, h, 2, 80, preset, True)
            for _ in range(20): fn()
            ts = []
            for _ in range(200):
                t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
            ts.sort(); row.append("preset %d %7.1f us (%d B)" % (preset, ts[len(ts) // 2] * 1e6, len(r)))
        print("%4dx%-4d %-8s: %s" % (w, h, kind, "   ".join(row)), flush=True)
