"""Per-piece wall times of the pipelined delivery (PIXO_HIP_DEBUG=trace): python tools/debug_pieces_timing.py"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch, synth
from pixo_amd import jpeg
w = h = int(os.environ.get('SIZE', '4096'))
px = synth.noise(w, h, 42)
d = torch.from_numpy(px).to("cuda:0"); torch.cuda.synchronize()
o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420).build()
pinned = torch.empty(w * h * 3 // 2 + (1 << 16), dtype=torch.uint8).pin_memory()
for i in range(4):
    sys.stderr.write("--- call %d\n" % i)
    t = time.perf_counter(); n = jpeg.encode_device_into(pinned, d, o); sys.stderr.write("total %.3f ms\n" % ((time.perf_counter() - t) * 1e3))
