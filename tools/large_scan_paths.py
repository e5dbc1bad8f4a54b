#!/usr/bin/env python3
"""Large scans (4096 groups of 192 blocks and more) from device pixels into a pinned buffer: the default path (a scan coded in PIECES: coefficient
kernel bands + scan_code + stuffing kernel per piece, the copy engine overlapped) against ONE piece (debug switch one_piece: the fused kernel storing
straight into the pinned buffer).  Wall microseconds per file, median of 21."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, synth
from pixo_amd import jpeg
for (w, h, ss) in ((4096, 4096, 0), (8192, 4096, 1), (8192, 8192, 1)):
    buf = torch.empty(w * h * 3 // (2 if ss else 1) + (1 << 16), dtype=torch.uint8).pin_memory()
    for opt in (False, True):
        o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling(ss)).optimize_huffman(opt).build()
        for kind in ("noise", "photo", "gradient"):
            px = synth.noise(w, h, 42) if kind == "noise" else (synth.photo(w, h, 42) if kind == "photo" else synth.gradient_rgb(w, h))
            d = torch.from_numpy(np.ascontiguousarray(px)).cuda()
            row = []
            for sw in (None, "one_piece"):
                jpeg.debug_configure(sw)
                for _ in range(3):
                    nb = jpeg.encode_device_into(buf, d, o)
                ts = []
                for _ in range(21):
                    t = time.perf_counter(); nb = jpeg.encode_device_into(buf, d, o); ts.append((time.perf_counter() - t) * 1e6)
                row.append("%s %.0f us" % ("pieces" if sw is None else "one piece", statistics.median(ts)))
            jpeg.debug_configure(None)
            print("%dx%d %s %s%s (%d bytes):" % (w, h, "4:2:0" if ss else "4:4:4", kind, " optimised tables" if opt else "", nb), " | ".join(row), "| fallbacks", jpeg.lookback_fallbacks())
            del d
