import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, synth
from pixo_amd import jpeg
w = h = 4096
px = synth.noise(w, h, 42); d = torch.from_numpy(px).to("cuda:0"); torch.cuda.synchronize()
for name, kw in [("baseline", {}), ("progressive", dict(progressive=True)), ("progressive+trellis", dict(progressive=True, trellis_quant=True))]:
    b = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420)
    for k, v in kw.items(): b = getattr(b, k)(v)
    o = b.build()
    pin = torch.zeros(w * h, dtype=torch.uint8).pin_memory()
    for fn_name, fn in (("encode_device", lambda: jpeg.encode_device(d, o)), ("encode_device_into pinned", lambda: jpeg.encode_device_into(pin, d, o))):
        fn(); ts = []
        for _ in range(7):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        print("%-22s %-26s median %.3f ms min %.3f ms" % (name, fn_name, sorted(ts)[3] * 1e3, min(ts) * 1e3), flush=True)
