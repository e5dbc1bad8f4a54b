"""End-to-end timings through the host-pointer C ABI (PCIe and host entropy stage included);
these are NOT bench.py's `value` — they go into DESIGN.md as the PCIe-inclusive rates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import synth
from pixo_amd import ColorType, jpeg
w = h = 4096
px = synth.noise(w, h, 42)
o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420).build()
for name, fn in [("coefficients() host->host (H2D + kernel + D2H + memcpy)", lambda: jpeg.coefficients(px, o)),
                 ("encode() whole file (H2D + kernels incl. device entropy stage + file D2H)", lambda: jpeg.encode(px, o))]:
    fn()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    print("%-70s %8.2f ms  %8.1f Mpixels/s" % (name, t * 1e3, w * h / t / 1e6))
y, cb, cr = jpeg.coefficients(px, o)
t0 = time.perf_counter(); blob = jpeg.entropy_encode(y, cb, cr, o); t = time.perf_counter() - t0
print("%-70s %8.2f ms  %8.1f Mpixels/s  (%d bytes)" % ("entropy_encode() host stage alone, 1 thread", t * 1e3, w * h / t / 1e6, len(blob)))
