"""Timing of the Bigrams strategy per shape (debug aid)."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import numpy as np
import synth
from pixo_amd import png
for (w, h, bpp) in [(256, 40, 4), (20000, 34, 4), (2, 2100, 1), (1, 5000, 1), (700, 40, 3), (4096, 64, 4), (4096, 4096, 4), (4100, 33, 6), (5, 900, 3)]:
    px = synth.lcg_bytes(w * h * bpp, 1)
    for s in (png.FilterStrategy.ADAPTIVE, png.FilterStrategy.BIGRAMS):
        png.apply_filters(px, w, h, bpp, s)
        t = time.perf_counter(); png.apply_filters(px, w, h, bpp, s); dt = time.perf_counter() - t
        print("%6dx%-5d bpp %d strategy %d: %9.3f ms" % (w, h, bpp, int(s), dt * 1e3), flush=True)
