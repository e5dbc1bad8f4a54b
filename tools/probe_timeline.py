#!/usr/bin/env python3
"""Timeline of ONE dispatch of the coefficient kernel from per-wavefront time stamps (timeline build of the library:
tools/ab_build.sh probe "-DPIXO_PROBE", selected with PIXO_HIP_LIB=pixo_amd/ab_probe.so; never the shipped library).

rocprofv3's thread trace (--att) cannot be decoded in this image (no rocprof-trace-decoder library), so the kernel
stamps the 100 MHz constant clock (s_memrealtime, one counter for all XCDs, 10 ns resolution) at eight points per
wavefront:  0 wavefront runs | 1 first work item arrived + converted | 2 last item converted | 3 barrier passed |
4 transform done | 5 quantised (first store next) | 6 last store issued | 7 stores acknowledged.

    PIXO_HIP_LIB=$PWD/pixo_amd/ab_probe.so python tools/probe_timeline.py [workload] [extra kernel variant label]
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from pixo_amd import _lib, jpeg  # noqa: E402

W, H, SS, BATCH = {"c2": (4096, 4096, 1, 1), "c2_444": (4096, 4096, 0, 1), "c3": (1920, 1080, 1, 64)}[sys.argv[1] if len(sys.argv) > 1 else "c2"]
label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.environ.get("PIXO_HIP_LIB", "libpixo_hip.so"))
dev = torch.device("cuda", 0)
L = _lib.load()
probe_set = L.pixo_hip_debug_probe_buffer
probe_set.argtypes = [C.c_void_p]
yb, cbn = jpeg.coefficient_geometry(W, H, 2, SS)
nbuf = max(2, -(-(640 << 20) // (W * H * 3 * BATCH * (2 if SS else 3))))
base = torch.from_numpy(np.ascontiguousarray(synth.noise(W, H, 42))).to(dev)
ins, outs = [], []
for i in range(nbuf):
    t = base.repeat(BATCH) if BATCH > 1 else base
    ins.append((t ^ torch.tensor(i & 0xFF, dtype=torch.uint8, device=dev)).contiguous() if i else t.contiguous())
    outs.append(tuple(torch.empty((BATCH * n, 64), dtype=torch.int16, device=dev) for n in (yb, cbn, cbn)))
tile_h = 16 if SS else 8
wgs = -(-W // 512) * -(-H // tile_h) * BATCH
buf = torch.zeros(wgs * 3 * 8, dtype=torch.int64, device=dev)
assert probe_set(buf.data_ptr()) == 0
stream = torch.cuda.current_stream().cuda_stream


def step(i):
    y, cb, cr = outs[i % nbuf]
    jpeg.coefficients_device(ins[i % nbuf], W, H, 2, SS, 80, y, cb, cr, batch=BATCH, stream=stream)


t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 0.15:  # steady clocks
    for _ in range(16):
        step(n); n += 1
    torch.cuda.synchronize()
runs = []
for rep in range(5):
    for _ in range(8):  # back to back like the benchmark; the stamps of the LAST launch survive
        step(n); n += 1
    torch.cuda.synchronize()
    runs.append(buf.cpu().numpy().reshape(wgs, 3, 8).astype(np.int64).copy())
print("timeline of one dispatch: %s, %dx%d x%d, %s, %d workgroups x 3 wavefronts, library %s" % (sys.argv[1] if len(sys.argv) > 1 else "c2", W, H, BATCH, "4:2:0" if SS else "4:4:4", wgs, label))
print("(us after the first wavefront's start; 100 MHz clock: 0.01 us resolution; five dispatches, each the last of eight back to back)")
names = ["wavefront runs", "first item arrived + converted", "last item converted", "barrier passed", "transform done", "quantised, first store next", "last store issued",
         "stores acknowledged"]
for k, s in enumerate(runs):
    t = (s - s[:, :, 0].min()) / 100.0
    if k == 0:
        print("%-32s %8s %8s %8s %8s %8s" % ("stamp", "min", "p10", "median", "p90", "max"))
        for j in range(8):
            x = t[:, :, j].ravel()
            print("%-32s %8.2f %8.2f %8.2f %8.2f %8.2f" % (names[j], x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max()))
        print("waves per microsecond bin:  waiting for pixels | colour | transform | quantise | storing | done")
        for b in range(0, int(t.max()) + 1):
            lo = b + 0.5
            wait = ((t[:, :, 0] <= lo) & (t[:, :, 1] > lo)).sum()
            col = ((t[:, :, 1] <= lo) & (t[:, :, 3] > lo)).sum()
            tr = ((t[:, :, 3] <= lo) & (t[:, :, 4] > lo)).sum()
            qu = ((t[:, :, 4] <= lo) & (t[:, :, 5] > lo)).sum()
            st = ((t[:, :, 5] <= lo) & (t[:, :, 7] > lo)).sum()
            dn = (t[:, :, 7] <= lo).sum()
            print("  t = %4.1f us   %5d %5d %5d %5d %5d %5d" % (lo, wait, col, tr, qu, st, dn))
    print("dispatch %d: last wavefront starts %.2f | first pixels converted %.2f | last pixels converted %.2f | first store %.2f | last store issued %.2f | last store acknowledged %.2f us"
          % (k, t[:, :, 0].max(), t[:, :, 1].min(), t[:, :, 2].max(), t[:, :, 5].min(), t[:, :, 6].max(), t[:, :, 7].max()))
probe_set(None)
