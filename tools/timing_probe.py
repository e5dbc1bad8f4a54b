"""PIXO_TIMING build probe: shader clock during the kernel and workgroup lifetimes, from
s_memtime stamps at workgroup start / end (calibration: profiles/r01_ubench_memtime.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
dbg = torch.zeros(8192, dtype=torch.int64, device="cuda:0")
os.environ["PIXO_DBG_PTR"] = str(dbg.data_ptr())
import synth
from pixo_amd import jpeg
w = h = 4096
px = torch.from_numpy(synth.noise(w, h, 1)).to("cuda:0")
yb, cbn = jpeg.coefficient_geometry(w, h, 2, 1)
y = torch.empty((yb, 64), dtype=torch.int16, device="cuda:0"); cb = torch.empty((cbn, 64), dtype=torch.int16, device="cuda:0"); cr = torch.empty_like(cb)
s = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    jpeg.coefficients_device(px, w, h, 2, 1, 80, y, cb, cr, stream=s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); jpeg.coefficients_device(px, w, h, 2, 1, 80, y, cb, cr, stream=s); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
d = dbg.cpu().numpy()[:4096].reshape(-1, 2)
life = (d[:, 1] - d[:, 0])
# s_memtime is per XCD (block b runs on XCD b % 8): compare stamps only within one XCD
for x in range(8):
    dx = d[x::8]
    span = int(dx[:, 1].max() - dx[:, 0].min())
    st = np.sort(dx[:, 0]) - dx[:, 0].min()
    en = np.sort(dx[:, 1]) - dx[:, 0].min()
    print("XCD %d: kernel %.1f us, span %d ticks => %.2f GHz; starts at [#0 %d, #96 %d, #191 %d, #192 %d, #255 %d]; first end %d"
          % (x, us, span, span / us / 1e3, st[0], st[96], st[191], st[192], st[255], en[0]))
print("workgroup lifetime ticks: mean %.0f min %d max %d" % (life.mean(), life.min(), life.max()))
gen1 = life[:1536]; gen2 = life[1536:]
print("first-generation lifetime mean %.0f, second-generation %.0f" % (gen1.mean(), gen2.mean()))
