"""Host-pointer PNG filter entry (pixo_hip_png_filter): wall time per call, 4096x4096 RGBA and smaller shapes."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import synth
from pixo_amd import png
for (w, h, bpp) in [(4096, 4096, 4), (1920, 1080, 3), (8192, 2048, 4)]:
    px = synth.lcg_bytes(w * h * bpp, 1)
    for s in (png.FilterStrategy.ADAPTIVE, png.FilterStrategy.ADAPTIVE_FAST, png.FilterStrategy.SUB):
        png.apply_filters(px, w, h, bpp, s)
        ts = []
        for _ in range(7):
            t = time.perf_counter(); out, ad = png.apply_filters(px, w, h, bpp, s); ts.append(time.perf_counter() - t)
        print("%5dx%-5d bpp %d %-14s median %8.3f ms  min %8.3f ms  (%.1f MB in, %.1f MB out)" % (w, h, bpp, s.name, sorted(ts)[3] * 1e3, min(ts) * 1e3, px.size / 1e6, out.size / 1e6), flush=True)
