"""One-off: repeated timing of one large progressive file (which kernel takes the time: run under `call.sh kstats`).
   python tools/prog_large_time.py [size] [kind] [subsampling 0|1] [reps]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import synth
from pixo_amd import jpeg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
kind = sys.argv[2] if len(sys.argv) > 2 else "gradient"
ss = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
px = synth.noise(n, n, 42) if kind == "noise" else (synth.gradient_rgb(n, n) if kind == "gradient" else synth.constant(n, n, 77))
d = torch.from_numpy(px).to("cuda:0"); torch.cuda.synchronize()
o = jpeg.JpegOptions.builder(n, n).quality(80).subsampling(jpeg.Subsampling(ss)).progressive(True).build()
for r in range(reps):
    t0 = time.perf_counter(); f = jpeg.encode_device(d, o); t = time.perf_counter() - t0
    print("%dx%d %s ss=%d rep %d: %d bytes, %.2f ms, fallbacks %d" % (n, n, kind, ss, r, len(f), t * 1e3, jpeg.lookback_fallbacks()), flush=True)
