#!/usr/bin/env python3
"""Device time per file of 4096-wide images of growing height (1, 2, 4 generations of the fused kernel's workgroups), fused and two-kernel form:
does a launch of several generations cost more per group than one generation?   python tools/device_time_tall.py [kind]"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, synth
from pixo_amd import jpeg
kind = sys.argv[1] if len(sys.argv) > 1 else "gradient"
stream = torch.cuda.current_stream().cuda_stream
for H in (2048, 4096, 8192, 16384):
    W = 4096
    O = jpeg.JpegOptions.builder(W, H).quality(80).subsampling(jpeg.Subsampling.S420).build()
    px = synth.noise(W, H, 42) if kind == "noise" else (synth.photo(W, H, 42) if kind == "photo" else synth.gradient_rgb(W, H))
    d = torch.from_numpy(np.ascontiguousarray(px)).cuda()
    row = []
    for sw in (None, "two_kernel_scan"):
        jpeg.debug_configure(sw)
        form = jpeg.debug_scan_device_async(d, O, stream=stream)
        torch.cuda.synchronize()
        for i in range(20):
            jpeg.debug_scan_device_async(d, O, stream=stream)
        torch.cuda.synchronize()
        evs = []
        for b in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20):
                jpeg.debug_scan_device_async(d, O, stream=stream)
            e1.record(); torch.cuda.synchronize()
            evs.append(e0.elapsed_time(e1) / 20 * 1e3)
        row.append("%s %.1f us" % ("fused" if form else "two-kernel", statistics.median(evs)))
    jpeg.debug_configure(None)
    print(kind, "%dx%d (%d groups):" % (W, H, (W // 512) * (H // 16)), " | ".join(row), "| fallbacks", jpeg.lookback_fallbacks())
