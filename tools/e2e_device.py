"""Whole-file timings for device-resident pixels (encode_device) and for a device-resident
coefficient tuple (entropy_encode_device); compare with tools/e2e_timing.py (host pointers)."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import synth
from pixo_amd import jpeg
w = h = 4096
px = synth.noise(w, h, 42)
o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420).build()
d_px = torch.from_numpy(px).to("cuda:0"); torch.cuda.synchronize()
def timeit(name, fn, n=7):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    print("%-60s median %8.3f ms  min %8.3f ms  (%s)" % (name, sorted(ts)[n // 2] * 1e3, min(ts) * 1e3, len(r) if hasattr(r, "__len__") else ""))
timeit("encode(host pixels)", lambda: jpeg.encode(px, o))
timeit("encode_device(device pixels)", lambda: jpeg.encode_device(d_px, o))
yb, cbn = jpeg.coefficient_geometry(w, h, 2, 1)
d_y = torch.empty((yb, 64), dtype=torch.int16, device="cuda:0"); d_cb = torch.empty((cbn, 64), dtype=torch.int16, device="cuda:0"); d_cr = torch.empty_like(d_cb)
jpeg.coefficients_device(d_px, w, h, 2, 1, 80, d_y, d_cb, d_cr); torch.cuda.synchronize()
timeit("entropy_encode_device(device tuple)", lambda: jpeg.entropy_encode_device(d_y, d_cb, d_cr, o))
g = synth.gradient_rgb(w, h); d_g = torch.from_numpy(g).to("cuda:0"); torch.cuda.synchronize()
timeit("encode_device(gradient image: 0.3 MB file)", lambda: jpeg.encode_device(d_g, o))
# config 3 shape: 64 x 1920x1080, whole files, one batched call vs one call per image
w3, h3, n3 = 1920, 1080, 64
o3 = jpeg.JpegOptions.builder(w3, h3).quality(80).subsampling(jpeg.Subsampling.S420).build()
imgs = torch.cat([torch.from_numpy(synth.noise(w3, h3, 42 + i)) for i in range(n3)]).to("cuda:0"); torch.cuda.synchronize()
timeit("encode_batch_device(64 x 1080p noise) per batch", lambda: jpeg.encode_batch_device(imgs, o3, n3), n=5)
per = w3 * h3 * 3
timeit("64 x encode_device(1080p noise), one call per image", lambda: [jpeg.encode_device(imgs[i * per:(i + 1) * per], o3) for i in range(n3)], n=3)
gimgs = torch.from_numpy(synth.gradient_rgb(w3, h3)).to("cuda:0").repeat(n3); torch.cuda.synchronize()
timeit("encode_batch_device(64 x 1080p gradient) per batch", lambda: jpeg.encode_batch_device(gimgs, o3, n3), n=5)
timeit("64 x encode_device(1080p gradient), one call per image", lambda: [jpeg.encode_device(gimgs[i * per:(i + 1) * per], o3) for i in range(n3)], n=3)
