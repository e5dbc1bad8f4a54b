"""Preset 2 (optimised tables + progressive + trellis) timings through encode_device, and the share
of the trellis kernel (rocprofv3 gives the kernel split)."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import synth
from pixo_amd import jpeg
for (w, h, kind) in [(1920, 1080, "noise"), (1920, 1080, "gradient"), (4096, 4096, "noise")]:
    px = synth.noise(w, h, 42) if kind == "noise" else synth.gradient_rgb(w, h)
    d = torch.from_numpy(px).to("cuda:0"); torch.cuda.synchronize()
    for name, kw in [("progressive", dict(progressive=True)), ("progressive+trellis", dict(progressive=True, trellis_quant=True)),
                     ("preset 2 (max)", dict(progressive=True, trellis_quant=True, optimize_huffman=True))]:
        b = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420)
        for k, v in kw.items(): b = getattr(b, k)(v)
        o = b.build()
        jpeg.encode_device(d, o)
        t0 = time.perf_counter(); n = 3
        for _ in range(n): blob = jpeg.encode_device(d, o)
        t = (time.perf_counter() - t0) / n
        print("%dx%d %-8s %-20s %9.2f ms  %8.1f Mpixels/s  %d bytes" % (w, h, kind, name, t * 1e3, w * h / t / 1e6, len(blob)))
