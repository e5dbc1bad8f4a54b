#!/usr/bin/env python3
"""Whole 4096x4096 files from DEVICE pixels into each thread's own pinned buffer, T calling threads at once (every thread has its own
context and stream inside the library): wall microseconds per file.
    python tools/mt_device_files.py [kind ...] [--sw debug,switches]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import synth
from pixo_amd import jpeg

W = H = 4096
O = jpeg.JpegOptions.builder(W, H).quality(80).subsampling(jpeg.Subsampling.S420).build()
args = sys.argv[1:]
sw = None
if "--sw" in args:
    i = args.index("--sw"); sw = args[i + 1]; del args[i:i + 2]
kinds = args or ["noise", "photo", "gradient"]
if sw:
    jpeg.debug_configure(sw)
for kind in kinds:
    px = synth.noise(W, H, 42) if kind == "noise" else (synth.photo(W, H, 42) if kind == "photo" else synth.gradient_rgb(W, H))
    d = torch.from_numpy(np.ascontiguousarray(px)).cuda()
    row = []
    for T in [int(x) for x in os.environ.get('TS', '1,2,3,4,8').split(',')]:
        bufs = [torch.empty(W * H * 3 // 2 + (1 << 16), dtype=torch.uint8).pin_memory() for _ in range(T)]
        gate = threading.Barrier(T + 1)
        N = 24
        calls = []

        def work(buf):
            jpeg.encode_device_into(buf, d, O)
            jpeg.encode_device_into(buf, d, O)
            gate.wait()
            for _ in range(N):
                t0 = time.perf_counter()
                jpeg.encode_device_into(buf, d, O)
                calls.append((time.perf_counter() - t0) * 1e6)
        ths = [threading.Thread(target=work, args=(b,)) for b in bufs]
        for t in ths:
            t.start()
        gate.wait()
        t1 = time.perf_counter()
        for t in ths:
            t.join()
        cs = sorted(calls)
        row.append("T=%d %.1f (fb %d; calls median %.0f, three slowest %s)" % (T, (time.perf_counter() - t1) / (N * T) * 1e6, jpeg.lookback_fallbacks(), cs[len(cs) // 2], " ".join("%.0f" % x for x in cs[-3:])))
        del bufs
    print(kind, "switches", sw, "| us per file:", "  ".join(row))
