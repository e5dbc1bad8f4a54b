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
# the same entry with the caller's result array REUSED (resident pages): what the library itself needs
import ctypes as C
from pixo_amd import _lib
L = _lib.load()
for (w, h, bpp) in [(4096, 4096, 4), (8192, 2048, 4)]:
    px = synth.lcg_bytes(w * h * bpp, 1)
    out = np.empty(png.filtered_size(w, h, bpp), np.uint8); ad = C.c_uint32()
    for s in (png.FilterStrategy.ADAPTIVE, png.FilterStrategy.SUB):
        ts = []
        for _ in range(9):
            t = time.perf_counter()
            rc = L.pixo_hip_png_filter(px.ctypes.data, px.size, w, h, bpp, int(s), 0, out.ctypes.data, out.size, C.byref(ad))
            ts.append(time.perf_counter() - t); assert rc == 0
        print("%5dx%-5d bpp %d %-14s result array reused: median %8.3f ms  min %8.3f ms" % (w, h, bpp, s.name, sorted(ts)[4] * 1e3, min(ts) * 1e3), flush=True)
