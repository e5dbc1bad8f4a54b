"""N encodes of ONE device-resident 4096x4096 image into pinned storage, for one option set — the loop behind the rocprofv3
dispatch counts of a progressive / preset-2 file (<= 12 dispatches, no __amd_rocclr_copyBuffer in steady
state).   python tools/preset2_probe.py <progressive|trellis|preset2|baseline> [n] [noise|gradient|photo]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import synth
from pixo_amd import jpeg
what = sys.argv[1] if len(sys.argv) > 1 else "progressive"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
kind = sys.argv[3] if len(sys.argv) > 3 else "noise"
w = h = 4096
px = synth.noise(w, h, 42) if kind == "noise" else (synth.gradient_rgb(w, h) if kind == "gradient" else (synth.photo(w, h, 42) if kind == "photo" else synth.constant(w, h, 77)))
d = torch.from_numpy(px).to("cuda:0"); torch.cuda.synchronize()
b = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420)
kw = {"baseline": {}, "progressive": dict(progressive=True), "trellis": dict(progressive=True, trellis_quant=True),
      "preset2": dict(progressive=True, trellis_quant=True, optimize_huffman=True)}[what]
for k, v in kw.items(): b = getattr(b, k)(v)
o = b.build()
pinned = torch.empty(w * h * 2, dtype=torch.uint8).pin_memory()  # (>= 64 B per block + headers: the library then may deliver in pieces)
for _ in range(3): jpeg.encode_device_into(pinned, d, o)
ts = []
for _ in range(n):
    t0 = time.perf_counter()
    nbytes = jpeg.encode_device_into(pinned, d, o)
    ts.append(time.perf_counter() - t0)
ts.sort()
print("%s 4096x4096 %s -> pinned: median %.3f ms, min %.3f, max %.3f over %d files, %d bytes" % (what, kind, ts[n // 2] * 1e3, ts[0] * 1e3, ts[-1] * 1e3, n, nbytes))
