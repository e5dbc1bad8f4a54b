#!/usr/bin/env python3
"""Device time of configs[2] — 64 x 1920x1080 RGB8, q=80, 4:2:0, device resident — through pixo_hip_debug_scan_device_async_batch (the
product's kernels of the one-pass batch, no waits, no PCIe): the fused pixel -> scan kernel with every image a segment against
coefficient kernel + scan_code + stuffing kernel (debug switch two_kernel_scan; the fused form is the default for batches of images with well-filled tiles, fused_batch forces it) in the same process; HIP events on the launch
stream, median of 7 blocks of 10 batches.    python tools/device_time_batch.py [batch] [w] [h]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import synth
from pixo_amd import jpeg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
O = jpeg.JpegOptions.builder(W, H).quality(80).subsampling(jpeg.Subsampling.S420).build()
stream = torch.cuda.current_stream().cuda_stream
for kind in ("noise", "photo", "gradient"):
    imgs = [synth.noise(W, H, 42 + i) if kind == "noise" else (synth.photo(W, H, 42 + i) if kind == "photo" else synth.gradient_rgb(W, H)) for i in range(min(N, 8))]
    one = np.concatenate([np.ascontiguousarray(imgs[i % len(imgs)]).reshape(-1) for i in range(N)])
    d = torch.from_numpy(one).cuda()
    row = []
    for form in ("fused", "two"):
        jpeg.debug_configure("two_kernel_scan" if form == "two" else "fused_batch")
        f = jpeg.debug_scan_device_async(d, O, stream=stream, batch=N)
        torch.cuda.synchronize()
        for _ in range(6):
            jpeg.debug_scan_device_async(d, O, stream=stream, batch=N)
        torch.cuda.synchronize()
        evs = []
        for b in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(10):
                jpeg.debug_scan_device_async(d, O, stream=stream, batch=N)
            e1.record()
            torch.cuda.synchronize()
            evs.append(e0.elapsed_time(e1) / 10 * 1e3)
        row.append("%s(form %d) %.1f us (min %.1f)" % (form, f, statistics.median(evs), min(evs)))
    jpeg.debug_configure(None)
    print("%d x %dx%d %s | %s | fallbacks %d" % (N, W, H, kind, " | ".join(row), jpeg.lookback_fallbacks()))
    del d
