#!/usr/bin/env python3
"""Times encode_device right after torch kernels on the producer stream (the scenario of
test_device_pixels_are_read_after_their_producer), call by call."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth
from pixo_amd import jpeg
w = h = 4096
px = synth.noise(w, h, 42)
o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420).build()
src = torch.from_numpy(px).to("cuda:0")
big = torch.empty(64 << 20, dtype=torch.float32, device="cuda:0")
jpeg.set_producer_stream(None)
def t(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
    print("%-50s %9.3f ms" % (label, dt * 1e3), flush=True); return r
t("encode_device (idle GPU)", lambda: jpeg.encode_device(src, o))
t("encode_device (idle GPU)", lambda: jpeg.encode_device(src, o))
for i in range(3):
    d_px = torch.zeros_like(src)
    t0 = time.perf_counter(); big.normal_(); d_px.copy_(src); t1 = time.perf_counter()
    r = jpeg.encode_device(d_px, o); t2 = time.perf_counter()
    print("normal_+copy enqueue %.3f ms, encode_device behind them %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
t("big.normal_() alone", lambda: (big.normal_(), torch.cuda.synchronize()))
t("encode_device (idle GPU)", lambda: jpeg.encode_device(src, o))
