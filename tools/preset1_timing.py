#!/usr/bin/env python3
"""Whole files with OPTIMISED tables (the reference's Balanced preset: optimize_huffman, 4:4:4; and the same at 4:2:0) from device pixels into a
pinned buffer: statistics from the pixels + the fused kernel (default) against coefficient kernel + scan_count + scan_code + stuffing kernel
(debug switch two_kernel_scan).  Wall microseconds per file, median of 31.   python tools/preset1_timing.py [size]"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, synth
from pixo_amd import jpeg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
buf = torch.empty(n * n * 3 + (1 << 16), dtype=torch.uint8).pin_memory()
for ss in (1, 0):
    o = jpeg.JpegOptions.builder(n, n).quality(80).subsampling(jpeg.Subsampling(ss)).optimize_huffman(True).build()
    for kind in ("noise", "photo", "gradient"):
        px = synth.noise(n, n, 42) if kind == "noise" else (synth.photo(n, n, 42) if kind == "photo" else synth.gradient_rgb(n, n))
        d = torch.from_numpy(np.ascontiguousarray(px)).cuda()
        row = []
        for sw in (None, "two_kernel_scan"):
            jpeg.debug_configure(sw)
            for _ in range(4):
                nb = jpeg.encode_device_into(buf, d, o)
            ts = []
            for _ in range(31):
                t = time.perf_counter(); nb = jpeg.encode_device_into(buf, d, o); ts.append((time.perf_counter() - t) * 1e6)
            row.append("%s %.0f us" % ("fused" if sw is None else "two-kernel", statistics.median(ts)))
        jpeg.debug_configure(None)
        print("%dx%d %s %s (%d bytes):" % (n, n, "4:2:0" if ss else "4:4:4", kind, nb), " | ".join(row), "| fallbacks", jpeg.lookback_fallbacks())
