"""One-off: the coefficient kernel on GRAY input (colour type 0: 1 B/px in, 2 B/px out) — time per launch by events,
rotating over buffer sets beyond the Infinity Cache, against the RGB 4:2:0 launch of the same size.
   python tools/gray_probe.py [size]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import synth
from pixo_amd import jpeg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
for ct, ss, name in ((0, 0, "gray"), (2, 1, "rgb 4:2:0"), (2, 0, "rgb 4:4:4")):
    ch = 1 if ct == 0 else 3
    yb, cbn = jpeg.coefficient_geometry(n, n, ct, ss)
    per = n * n * ch + (yb + 2 * cbn) * 128
    nbuf = max(2, -(-(640 << 20) // per))
    base = synth.noise_gray(n, n, 42) if ct == 0 else synth.noise(n, n, 42)
    ins = [torch.from_numpy(base).to(dev) ^ torch.tensor(i, dtype=torch.uint8, device=dev) for i in range(nbuf)]
    outs = [(torch.empty((yb, 64), dtype=torch.int16, device=dev), torch.empty((max(cbn, 1), 64), dtype=torch.int16, device=dev),
             torch.empty((max(cbn, 1), 64), dtype=torch.int16, device=dev)) for _ in range(nbuf)]
    def step(i):
        k = i % nbuf
        jpeg.coefficients_device(ins[k], n, n, ct, ss, 80, outs[k][0], outs[k][1], outs[k][2], batch=1, stream=stream)
    for i in range(3000): step(i)
    torch.cuda.synchronize()
    best = []
    for rep in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(500): step(i)
        b.record(); torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / 500 * 1e3)
    best.sort()
    t = best[len(best) // 2]
    print("%dx%d %-10s: %.2f us per launch (min %.2f), algorithmic %.1f MB -> %.2f TB/s = %.3f of 8 TB/s" % (n, n, name, t, best[0], per / 1e6, per / t / 1e6, per / t / 1e6 / 8))
