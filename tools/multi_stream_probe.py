#!/usr/bin/env python3
"""C2 kernel launched round-robin on S HIP streams (independent images in flight): Mpixels/s per S."""
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
K = 400
for S in (1, 2, 3, 4, 1):
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    def step(i):
        k = i % nbuf
        jpeg.coefficients_device(ins[k], w, h, 2, ss, q, *outs[k], stream=streams[i % S].cuda_stream)
    for i in range(40): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K): step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d stream(s): %7.2f us per image  %9.0f Mpixels/s" % (S, dt / K * 1e6, w * h * K / dt / 1e6), flush=True)
