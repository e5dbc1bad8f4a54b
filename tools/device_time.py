#!/usr/bin/env python3
"""Device time per baseline file (pixo_hip_debug_scan_device_async: the product's kernels, no waits, no PCIe), 4096x4096 q=80
4:2:0, for noise / photo / gradient content; HIP events on the launch stream, median of 7 blocks of 50 files.
    [PIXO_HIP_LIB=tools/ab/ab_x.so] python tools/device_time.py [two]     (`two`: debug switch two_kernel_scan)"""
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

W = H = 4096
SS = jpeg.Subsampling.S444 if os.environ.get("SS") == "444" else jpeg.Subsampling.S420  # (SS=444: the reference's default subsampling)
Q = int(os.environ.get("Q", "80"))
GRAY = os.environ.get("SS") == "gray"  # (SS=gray: Gray8 pixels)
O = jpeg.JpegOptions.builder(W, H).quality(Q).subsampling(SS).build() if not GRAY else jpeg.JpegOptions.builder(W, H).color_type(jpeg.ColorType(0)).quality(Q).build()
if len(sys.argv) > 1 and sys.argv[1] == "two":
    jpeg.debug_configure("two_kernel_scan")
stream = torch.cuda.current_stream().cuda_stream
out = []
for kind in ("noise", "photo", "gradient"):
    px = synth.noise(W, H, 42) if kind == "noise" else (synth.photo(W, H, 42) if kind == "photo" else synth.gradient_rgb(W, H))
    if GRAY:
        px = px.reshape(-1, 3)[:, 1].copy()
    ds = [torch.from_numpy(np.ascontiguousarray(px)).cuda() for _ in range(4)]  # (rotating copies: 200 MB > one L2, < Infinity Cache; the kernels' traffic is what it is)
    form = jpeg.debug_scan_device_async(ds[0], O, stream=stream)
    torch.cuda.synchronize()
    for i in range(60):
        jpeg.debug_scan_device_async(ds[i % 4], O, stream=stream)
    torch.cuda.synchronize()
    evs = []
    for b in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(50):
            jpeg.debug_scan_device_async(ds[i % 4], O, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        evs.append(e0.elapsed_time(e1) / 50 * 1e3)
    out.append("%s %.2f us (min %.2f)" % (kind, statistics.median(evs), min(evs)))
    del ds
print(os.environ.get("PIXO_HIP_LIB", "in-tree").split("/")[-1], "fused" if form else "two-kernel", "|", " | ".join(out), "| fallbacks", jpeg.lookback_fallbacks())
