"""Debug aid: device progressive scan coder vs its host twin, first differing byte and scan."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import synth
from pixo_amd import jpeg

def scans(b):
    out, i = [], 0
    while True:
        j = b.find(b"\xff\xda", i)
        if j < 0: break
        out.append(j); i = j + 2
    return out

for (w, h, preset_kw) in [(1024, 1024, 0), (2048, 2048, 0), (4096, 4096, 0), (4096, 4096, 1), (4096, 4096, 2)]:
    px = synth.noise(w, h, 77)
    b = jpeg.JpegOptions.builder(w, h).quality(85).subsampling(jpeg.Subsampling.S420).progressive(True)
    if preset_kw >= 1: b = b.optimize_huffman(True)
    if preset_kw >= 2: b = b.trellis_quant(True)
    o = b.build()
    jpeg.debug_configure("")
    dev = jpeg.encode(px, o)
    jpeg.debug_configure("host_entropy")
    host = jpeg.encode(px, o)
    jpeg.debug_configure(None)
    if dev == host:
        print(w, h, preset_kw, "equal", len(dev)); continue
    a, c = np.frombuffer(dev, np.uint8), np.frombuffer(host, np.uint8)
    m = min(len(a), len(c))
    d = np.nonzero(a[:m] != c[:m])[0]
    print(w, h, preset_kw, "DIFF len", len(dev), len(host), "first", d[0] if len(d) else None, "ndiff", len(d))
    print("  dev scans ", scans(dev)); print("  host scans", scans(host))
    if len(d):
        k = int(d[0]); print("  dev ", dev[k-8:k+8].hex(), "\n  host", host[k-8:k+8].hex())
