import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, synth
from pixo_amd import jpeg
w = h = 4096
px = synth.noise(w, h, 42)
o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420).build()
pin = torch.zeros(w * h * 2, dtype=torch.uint8).pin_memory().numpy()
for mode in sys.argv[1:]:
    jpeg.debug_configure(mode)
    jpeg.encode_into_buffer(pin, px, o); ts = []
    for _ in range(15):
        t0 = time.perf_counter(); n = jpeg.encode_into_buffer(pin, px, o); ts.append(time.perf_counter() - t0)
    print("%-60s median %.3f ms  min %.3f ms" % (mode or "(default)", sorted(ts)[7] * 1e3, min(ts) * 1e3), flush=True)
