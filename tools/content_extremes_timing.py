#!/usr/bin/env python3
"""Landmine hunt: wall ms per 4096x4096 file (device pixels -> pinned buffer, median of 7) for content at the extremes — flat, a slow ramp, sparse specks,
black / white noise, saturated noise at q = 100 — through every kind of file the library writes: baseline (4:2:0, 4:4:4, gray), optimised tables,
progressive, preset 2 (trellis + progressive + optimised).  Anything in the milliseconds that is not PCIe is a chain somewhere."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, synth
from pixo_amd import jpeg, ColorType
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N3 = n * n * 3
def contents():
    yield "flat", np.full(N3, 137, np.uint8)
    yield "slow ramp", ((np.arange(N3, dtype=np.int64) // 3 // 105) % 256).astype(np.uint8)
    b = np.full(N3, 90, np.uint8); k = synth.lcg_bytes(N3, 77); b[k < 2] = 165
    yield "sparse specks", b
    b = synth.lcg_bytes(N3, 78).copy(); b[b < 128] = 0; b[b >= 128] = 255
    yield "black/white noise", b
    yield "photo", synth.photo(n, n, 42)
buf = torch.empty(N3 + (1 << 16), dtype=torch.uint8).pin_memory()
kinds = [("4:2:0", dict(ss=1)), ("4:4:4", dict(ss=0)), ("gray", dict(gray=True)), ("4:2:0 optimised", dict(ss=1, opt=True)), ("4:2:0 progressive", dict(ss=1, prog=True)),
         ("preset 2", dict(ss=1, opt=True, prog=True, trellis=True)), ("4:2:0 q100", dict(ss=1, q=100)), ("4:2:0 restart rows", dict(ss=1, rst=2 * (n // 16)))]
for cname, px in contents():
    row = []
    for kname, k in kinds:
        gray = k.get("gray", False)
        p = px.reshape(-1, 3)[:, 1].copy() if gray else px
        d = torch.from_numpy(np.ascontiguousarray(p)).cuda()
        b = jpeg.JpegOptions.builder(n, n).color_type(ColorType(0 if gray else 2)).quality(k.get("q", 80)).subsampling(jpeg.Subsampling(k.get("ss", 0)))
        b = b.optimize_huffman(k.get("opt", False)).progressive(k.get("prog", False)).trellis_quant(k.get("trellis", False))
        if k.get("rst"): b = b.restart_interval(k["rst"])
        o = b.build()
        for _ in range(2): nb = jpeg.encode_device_into(buf, d, o)
        ts = []
        for _ in range(7):
            t = time.perf_counter(); nb = jpeg.encode_device_into(buf, d, o); ts.append((time.perf_counter() - t) * 1e3)
        row.append("%s %.3f" % (kname, statistics.median(ts)))
        del d
    print("%-18s" % cname, " | ".join(row), "| fallbacks", jpeg.lookback_fallbacks(), flush=True)
