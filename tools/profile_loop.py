"""Loops for counter profiles (rocprofv3 --pmc through tools/gpu/call.sh issue / traffic): 4096x4096 q=80 4:2:0 files from DEVICE pixels
into a pinned buffer.     python tools/profile_loop.py <noise|photo|gradient> [baseline|two|progressive|preset2] [n]
baseline: the fused pixel -> scan kernel; two: coefficient kernel + scan_code + stuffing kernel (debug switch two_kernel_scan);
progressive: prog_code + stuffing kernel; preset2: trellis + progressive + optimised tables."""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import torch
import synth
from pixo_amd import jpeg

kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
form = sys.argv[2] if len(sys.argv) > 2 else "baseline"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
w = h = 4096
px = synth.noise(w, h, 42) if kind == "noise" else (synth.photo(w, h, 42) if kind == "photo" else synth.gradient_rgb(w, h))
d = torch.from_numpy(np.ascontiguousarray(px)).cuda()
b = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420)
if form in ("progressive", "preset2"):
    b = b.progressive(True)
if form == "preset2":
    b = b.trellis_quant(True).optimize_huffman(True)
if form == "two":
    jpeg.debug_configure("two_kernel_scan")
o = b.build()
pinned = torch.empty(w * h * 3 // 2 + (1 << 16), dtype=torch.uint8).pin_memory()
nb = 0
for _ in range(n + 2):
    nb = jpeg.encode_device_into(pinned, d, o)
print("%s %s: %d files of %d bytes, fallbacks %d" % (kind, form, n, nb, jpeg.lookback_fallbacks()))
