"""One-off, under `rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats`: 200 encodes of one small image from host
pixels, so that the HIP calls of ONE encode can be counted and timed.   python tools/small_trace_probe.py [w] [h]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import synth
from pixo_amd import jpeg
w = int(sys.argv[1]) if len(sys.argv) > 1 else 64
h = int(sys.argv[2]) if len(sys.argv) > 2 else 64
px = synth.noise(w, h, 42)
o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling(1)).build()
for _ in range(200): jpeg.encode(px, o)
