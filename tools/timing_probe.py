"""PIXO_TIMING build probe: per-role cycle breakdown (s_memtime sums per workgroup)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
dbg = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda:0")
os.environ["PIXO_DBG_PTR"] = str(dbg.data_ptr())
import synth
from pixo_amd import jpeg
w = h = 4096
px = torch.from_numpy(synth.noise(w, h, 1)).to("cuda:0")
yb, cbn = jpeg.coefficient_geometry(w, h, 2, 1)
y = torch.empty((yb, 64), dtype=torch.int16, device="cuda:0"); cb = torch.empty((cbn, 64), dtype=torch.int16, device="cuda:0"); cr = torch.empty_like(cb)
for _ in range(3):
    jpeg.coefficients_device(px, w, h, 2, 1, 80, y, cb, cr, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16)
d = d[d[:, 0] != 0]
print("workgroups", len(d), "(cycles are s_memtime ticks summed over the workgroup's tiles; mean over workgroups)")
names = ["producer: wait prev loads", "producer: issue loads", "producer: convert", "producer: barrier"]
for i, n in enumerate(names): print("  %-28s %9.0f" % (n, d[:, i].mean()))
for wv in range(3):
    for i, n in enumerate(["barrier wait", "rows", "cols+quant", "store"]):
        print("  consumer %d %-18s %9.0f" % (wv, n, d[:, 4 + wv * 4 + i].mean()))
