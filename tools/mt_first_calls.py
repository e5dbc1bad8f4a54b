#!/usr/bin/env python3
"""First calls of T threads of a fresh process (every thread makes its context): per-call wall times and the look-back fallback count."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, synth
from pixo_amd import jpeg
W = H = 4096
O = jpeg.JpegOptions.builder(W, H).quality(80).subsampling(jpeg.Subsampling.S420).build()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kind = sys.argv[2] if len(sys.argv) > 2 else "noise"
px = synth.noise(W, H, 42) if kind == "noise" else synth.gradient_rgb(W, H)
d = torch.from_numpy(np.ascontiguousarray(px)).cuda()
bufs = [torch.empty(W * H * 3 // 2 + (1 << 16), dtype=torch.uint8).pin_memory() for _ in range(T)]
times = [[] for _ in range(T)]
fbs = [[] for _ in range(T)]
gate = threading.Barrier(T)
def work(i):
    gate.wait()
    for k in range(6):
        t = time.perf_counter(); jpeg.encode_device_into(bufs[i], d, O); times[i].append((time.perf_counter() - t) * 1e3); fbs[i].append(jpeg.lookback_fallbacks())
ths = [threading.Thread(target=work, args=(i,)) for i in range(T)]
[t.start() for t in ths]; [t.join() for t in ths]
print(os.environ.get("PIXO_HIP_LIB", "tree").split("/")[-1], "T", T, kind, "fallbacks", jpeg.lookback_fallbacks(), "| ms per call:", " / ".join(" ".join("%.2f" % x for x in r) for r in times), "| fb seen:", fbs[0])
