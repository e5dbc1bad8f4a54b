"""Kernel time of the coefficient stage for arbitrary shapes (device resident, HIP events, rotating buffers): 
python tools/shape_timing.py W H [W H ...]   (PIXO_HIP_LIB selects the build)"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import synth
from pixo_amd import jpeg
shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(3072, 4096)]
for (w, h) in shapes:
    base = torch.from_numpy(synth.noise(w, h, 42))
    nbuf = max(2, int(700e6 // (w * h * 3)))
    ins = [(base.to("cuda:0") ^ torch.tensor(i, dtype=torch.uint8, device="cuda:0")).contiguous() for i in range(min(nbuf, 9))]
    yb, cb = jpeg.coefficient_geometry(w, h, 2, 1)
    outs = [(torch.empty((yb, 64), dtype=torch.int16, device="cuda:0"), torch.empty((cb, 64), dtype=torch.int16, device="cuda:0"),
             torch.empty((cb, 64), dtype=torch.int16, device="cuda:0")) for _ in ins]
    s = torch.cuda.current_stream().cuda_stream
    def go(n):
        for i in range(n):
            k = i % len(ins)
            jpeg.coefficients_device(ins[k], w, h, 2, 1, 80, outs[k][0], outs[k][1], outs[k][2], 1, s)
    go(3000); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); go(200); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 200 * 1e3)
    us = sorted(ts)[3]
    print("%5dx%-5d  %6d workgroups  %7.2f us  %6.1f Gpx/s  frac %.3f" % (w, h, ((w + 511) // 512) * ((h + 15) // 16), us, w * h / us / 1e3, w * h * 6 / (us * 1e-6) / 8e12), flush=True)
