"""Whole-file encode() in a loop (for rocprofv3 --kernel-trace --stats of the device entropy stage).
usage: python tools/encode_loop.py [n] [optimize 0|1] [kind noise|gradient]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import synth
from pixo_amd import jpeg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
opt = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
kind = sys.argv[3] if len(sys.argv) > 3 else "noise"
w = h = 4096
px = synth.noise(w, h, 42) if kind == "noise" else synth.gradient_rgb(w, h)
o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420).optimize_huffman(opt).build()
jpeg.encode(px, o)
t0 = time.perf_counter()
for _ in range(n):
    blob = jpeg.encode(px, o)
t = (time.perf_counter() - t0) / n
print("encode() %s optimize=%d: %.2f ms per 4096x4096 image, %.0f Mpixels/s, %d bytes" % (kind, opt, t * 1e3, w * h / t / 1e6, len(blob)))
