"""Timing of the adaptive strategies on rows too long for the register path (two-pass form): 8192 x 2048 RGBA."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import numpy as np, torch
import synth
from pixo_amd import png
w, h, bpp = 8192, 2048, 4
base = torch.from_numpy(synth.lcg_bytes(w * h * bpp, 5))
ins = [(base.to("cuda:0") ^ torch.tensor(i, dtype=torch.uint8, device="cuda:0")).contiguous() for i in range(5)]
outs = [torch.empty(png.filtered_size(w, h, bpp), dtype=torch.uint8, device="cuda:0") for _ in range(5)]
sums = torch.zeros(2 * h, dtype=torch.int64, device="cuda:0"); scratch = torch.zeros(4, dtype=torch.int32, device="cuda:0")
s = torch.cuda.current_stream().cuda_stream
for name in ("PAETH", "ADAPTIVE_FAST", "ADAPTIVE"):
    st = getattr(png.FilterStrategy, name)
    for i in range(1500): png.apply_filters_async(ins[i % 5], w, h, bpp, outs[i % 5], sums, scratch, st, 0, s)
    torch.cuda.synchronize(); n = 300; t = time.perf_counter()
    for i in range(n): png.apply_filters_async(ins[i % 5], w, h, bpp, outs[i % 5], sums, scratch, st, 0, s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print("%-14s %7.1f us  %8.0f Mpixels/s" % (name, dt * 1e6, w * h / dt / 1e6), flush=True)
