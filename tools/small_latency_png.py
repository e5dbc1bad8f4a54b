"""One-off: latency of the PNG row-filter stage for one small RGBA image, host pixels -> filtered stream in host memory."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import synth
from pixo_amd import png
for (w, h) in ((64, 64), (256, 256), (512, 512), (1024, 1024), (1920, 1080)):
    px = synth.lcg_bytes(w * h * 4, 42)
    row = []
    for name, s in (("Adaptive", png.FilterStrategy.ADAPTIVE), ("AdaptiveFast", png.FilterStrategy.ADAPTIVE_FAST), ("Sub", png.FilterStrategy.SUB)):
        fn = lambda: png.apply_filters(px, w, h, 4, s, 0)
        for _ in range(20): fn()
        ts = []
        for _ in range(200):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        ts.sort(); row.append("%s %7.1f us" % (name, ts[len(ts) // 2] * 1e6))
    print("%4dx%-4d RGBA: %s" % (w, h, "   ".join(row)), flush=True)
