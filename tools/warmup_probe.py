#!/usr/bin/env python3
"""Time course of the C2 kernel rate after an idle period: consecutive windows of 200 launches."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import numpy as np, torch
import synth
from pixo_amd import jpeg
w = h = 4096; ss = 1; q = 80
dev = torch.device("cuda", 0)
yb, cbn = jpeg.coefficient_geometry(w, h, 2, ss)
base = torch.from_numpy(synth.noise(w, h, 42))
nbuf = 7
ins = [(base.to(dev) ^ torch.tensor(i, dtype=torch.uint8, device=dev)).contiguous() for i in range(nbuf)]
outs = [(torch.empty((yb, 64), dtype=torch.int16, device=dev), torch.empty((cbn, 64), dtype=torch.int16, device=dev),
         torch.empty((cbn, 64), dtype=torch.int16, device=dev)) for _ in range(nbuf)]
s = torch.cuda.current_stream().cuda_stream
def step(i):
    k = i % nbuf
    jpeg.coefficients_device(ins[k], w, h, 2, ss, q, *outs[k], stream=s)
for trial in range(2):
    torch.cuda.synchronize(); time.sleep(1.0)
    t_start = time.perf_counter()
    line = []
    for win in range(16):
        K = 200 if win < 12 else 2000
        t0 = time.perf_counter()
        for i in range(K): step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        line.append("%.1f" % (dt / K * 1e6))
    print("after 1 s idle, us per launch in consecutive windows (12 x 200, 4 x 2000 launches):", " ".join(line),
          " total %.0f ms" % ((time.perf_counter() - t_start) * 1e3), flush=True)
