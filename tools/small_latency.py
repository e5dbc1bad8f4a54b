"""One-off: latency of ONE small image (configs[0]'s shape 512x512 and neighbours), host pixels -> file and device pixels ->
pinned file, median over many calls.   python tools/small_latency.py"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import synth
from pixo_amd import jpeg
for (w, h) in ((64, 64), (256, 256), (512, 512), (1024, 1024), (1920, 1080)):
    for kind in ("noise", "gradient"):
        px = synth.noise(w, h, 42) if kind == "noise" else synth.gradient_rgb(w, h)
        o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling(1)).build()
        d = torch.from_numpy(px).to("cuda:0"); torch.cuda.synchronize()
        pinned = torch.empty(w * h * 2 + 65536, dtype=torch.uint8).pin_memory()
        res = []
        for fn in (lambda: jpeg.encode(px, o), lambda: jpeg.encode_device_into(pinned, d, o)):
            for _ in range(20): fn()
            ts = []
            for _ in range(300):
                t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
            ts.sort(); res.append(ts[len(ts) // 2] * 1e6)
        n = len(jpeg.encode(px, o))
        print("%4dx%-4d %-8s: host pixels -> bytes %7.1f us   device pixels -> pinned %7.1f us   (%d bytes)" % (w, h, kind, res[0], res[1], n), flush=True)
