"""Kernel time of the PNG filter stage per strategy (4096x4096 RGBA, device resident)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import synth
from pixo_amd import png
w = h = 4096; bpp = 4
base = torch.from_numpy(synth.rgba_noise_alpha1(w, h, 42))
nbuf = 5
ins = [(base.to("cuda:0") ^ torch.tensor(i, dtype=torch.uint8, device="cuda:0")).contiguous() for i in range(nbuf)]
outs = [torch.empty(png.filtered_size(w, h, bpp), dtype=torch.uint8, device="cuda:0") for _ in range(nbuf)]
sums = torch.zeros(2 * h, dtype=torch.int64, device="cuda:0"); scratch = torch.zeros(4, dtype=torch.int32, device="cuda:0")
s = torch.cuda.current_stream().cuda_stream
for name in ("NONE", "SUB", "UP", "AVERAGE", "PAETH", "ADAPTIVE_FAST", "ADAPTIVE", "BIGRAMS"):
    st = png.FilterStrategy[name]
    for i in range(5): png.apply_filters_async(ins[i % nbuf], w, h, bpp, outs[i % nbuf], sums, scratch, st, 0, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for i in range(n): png.apply_filters_async(ins[i % nbuf], w, h, bpp, outs[i % nbuf], sums, scratch, st, 0, s)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print("%-14s %8.1f us  %7.1f Gpx/s  %6.2f TB/s algorithmic" % (name, us, w * h / us / 1e3, (w * h * bpp * 2 + h) / us / 1e6))
