import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import synth, oracle_lib as O
from pixo_amd import jpeg
w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ss = int(sys.argv[4]) if len(sys.argv) > 4 else 1
imgs = [synth.noise(w, h, 42 + i) for i in range(n)]
yb, cbn = jpeg.coefficient_geometry(w, h, 2, ss)
dev = torch.device("cuda:0")
d_px = torch.from_numpy(np.concatenate(imgs)).to(dev)
d_y = torch.full((n * yb, 64), -7777, dtype=torch.int16, device=dev)
d_cb = torch.full((n * cbn, 64), -7777, dtype=torch.int16, device=dev)
d_cr = torch.full((n * cbn, 64), -7777, dtype=torch.int16, device=dev)
jpeg.coefficients_device(d_px, w, h, 2, ss, 80, d_y, d_cb, d_cr, batch=n, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
y, cb, cr = d_y.cpu().numpy(), d_cb.cpu().numpy(), d_cr.cpu().numpy()
unit = 16 if ss else 8
ux = (w + unit - 1) // unit
per = 4 if ss else 1
for i in range(n):
    oy, ocb, ocr = O.coeffs(imgs[i], w, h, 2, ss, 80, threads=8)
    by = np.where((y[i * yb:(i + 1) * yb] != oy).any(axis=1))[0]
    bc = np.where((cb[i * cbn:(i + 1) * cbn] != ocb).any(axis=1))[0]
    unw = (y[i * yb:(i + 1) * yb] == -7777).all(axis=1).sum()
    mc = sorted(set((b // per) for b in by))
    print("img", i, "bad Y blocks", len(by), "bad Cb", len(bc), "unwritten Y blocks", unw,
          "mcu (row,col) sample", [(m // ux, m % ux) for m in mc[:12]], "...", [(m // ux, m % ux) for m in mc[-4:]])
