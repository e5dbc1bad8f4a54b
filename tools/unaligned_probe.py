import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import synth
from pixo_amd import jpeg
for (w, h) in [(4096, 4096), (4094, 4096), (4095, 4095), (1921, 1080)]:
    px = torch.from_numpy(synth.noise(w, h, 1)).to("cuda:0")
    yb, cbn = jpeg.coefficient_geometry(w, h, 2, 1)
    y = torch.empty((yb, 64), dtype=torch.int16, device="cuda:0"); cb = torch.empty((cbn, 64), dtype=torch.int16, device="cuda:0"); cr = torch.empty_like(cb)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5): jpeg.coefficients_device(px, w, h, 2, 1, 80, y, cb, cr, stream=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): jpeg.coefficients_device(px, w, h, 2, 1, 80, y, cb, cr, stream=s)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("%dx%d: %.1f us, %.0f Mpx/s" % (w, h, us, w * h / us))
