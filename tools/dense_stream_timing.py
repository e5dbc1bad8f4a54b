#!/usr/bin/env python3
"""Files of more than 30 bytes per block (photographs at q = 100, noise at q >= 90): the context's last file sends the next one through the
two-kernel form (pieces.cpp dense_stream).  Wall us per 4096x4096 file, device pixels -> pinned buffer, median of 15 — default against the fused
kernel forced (debug switch fused_batch)."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, synth
from pixo_amd import jpeg
n = 4096
buf = torch.empty(n * n * 3 + (1 << 16), dtype=torch.uint8).pin_memory()
for q, kind in ((100, "photo"), (90, "noise"), (80, "photo")):
    px = synth.noise(n, n, 42) if kind == "noise" else synth.photo(n, n, 42)
    d = torch.from_numpy(np.ascontiguousarray(px)).cuda()
    o = jpeg.JpegOptions.builder(n, n).quality(q).subsampling(jpeg.Subsampling.S420).build()
    row = []
    for sw in (None, "fused_batch"):
        jpeg.debug_configure(sw)
        for _ in range(3): nb = jpeg.encode_device_into(buf, d, o)
        ts = []
        for _ in range(15):
            t = time.perf_counter(); nb = jpeg.encode_device_into(buf, d, o); ts.append((time.perf_counter() - t) * 1e6)
        row.append("%s %.0f us" % ("default" if sw is None else "fused forced", statistics.median(ts)))
    jpeg.debug_configure(None)
    print("q %d %s (%d bytes, %.1f B/block):" % (q, kind, nb, nb / (n * n / 64 * 1.5)), " | ".join(row))
