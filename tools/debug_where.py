import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle_lib as O
from pixo_amd import jpeg
w, h, n = 1920, 1080, 8
def img(i):
    x = np.arange(w)[None, :] // 8
    y = np.arange(h)[:, None] // 8
    a = np.zeros((h, w, 3), np.uint8)
    a[..., 0] = x % 256
    a[..., 1] = y % 256
    a[..., 2] = i * 30
    return a.reshape(-1)
imgs = [img(i) for i in range(n)]
yb, cbn = jpeg.coefficient_geometry(w, h, 2, 1)
dev = torch.device("cuda:0")
d_px = torch.from_numpy(np.concatenate(imgs)).to(dev)
d_y = torch.full((n * yb, 64), -7777, dtype=torch.int16, device=dev)
d_cb = torch.full((n * cbn, 64), -7777, dtype=torch.int16, device=dev)
d_cr = torch.full((n * cbn, 64), -7777, dtype=torch.int16, device=dev)
jpeg.coefficients_device(d_px, w, h, 2, 1, 100, d_y, d_cb, d_cr, batch=n, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
y = d_y.cpu().numpy().reshape(n, -1, 4, 64)
cb = d_cb.cpu().numpy().reshape(n, -1, 64)
# expected per image
table = {}
exp = []
for i in range(n):
    oy, ocb, ocr = O.coeffs(imgs[i], w, h, 2, 1, 100)
    oy = oy.reshape(-1, 4, 64)
    exp.append((oy, ocb))
    for m in range(oy.shape[0]):
        table[(int(oy[m, 0, 0]), int(oy[m, 1, 0]), int(oy[m, 2, 0]), int(ocb[m, 0]))] = (i, m // 120, m % 120)
for i in range(n):
    oy, ocb = exp[i]
    bad = np.where((y[i] != oy).any(axis=(1, 2)))[0]
    if len(bad) == 0:
        continue
    for m in list(bad[:3]) + list(bad[-2:]):
        key = (int(y[i, m, 0, 0]), int(y[i, m, 1, 0]), int(y[i, m, 2, 0]), int(cb[i, m, 0]))
        print("img", i, "mcu", (m // 120, m % 120), "got data of", table.get(key, key), "AC nonzero:", int(np.count_nonzero(y[i, m, :, 1:])))
