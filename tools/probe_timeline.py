#!/usr/bin/env python3
"""Timeline of ONE dispatch of the coefficient kernel from per-wavefront time stamps (timeline build of the library:
tools/ab_build.sh probe "-DPIXO_PROBE", selected with PIXO_HIP_LIB=tools/ab/ab_probe.so; never the shipped library).

rocprofv3's thread trace (--att) cannot be decoded in this image (no rocprof-trace-decoder library), so the kernel
stamps the 100 MHz constant clock (s_memrealtime, one counter for all XCDs, 10 ns resolution) at eight points per
wavefront:  0 wavefront runs | 1 first work item arrived + converted | 2 last item converted | 3 barrier passed |
4 transform done | 5 quantised (first store next) | 6 last store issued | 7 stores acknowledged.

    PIXO_HIP_LIB=$PWD/tools/ab/ab_probe.so python tools/probe_timeline.py [workload] [extra kernel variant label]
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
SLOTS, LAUNCHES = 12, 8
words = wgs * 3 * SLOTS
buf = torch.zeros(LAUNCHES * words, dtype=torch.int64, device=dev)
probe_set.argtypes = [C.c_void_p, C.c_uint]
assert probe_set(buf.data_ptr(), words) == 0
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
for _ in range(64):  # a multiple of LAUNCHES back to back: the last eight launches' stamps survive, in launch order
    step(n); n += 1
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(LAUNCHES, wgs, 3, SLOTS).astype(np.int64)
first = int(np.argmin(raw[:, :, :, 0].min(axis=(1, 2))))  # the part written by the oldest of the eight
order = [(first + k) % LAUNCHES for k in range(LAUNCHES)]
t00 = raw[order[0], :, :, 0].min()
print("timeline: %s, %dx%d x%d, %s, %d workgroups x 3 wavefronts, library %s" % (sys.argv[1] if len(sys.argv) > 1 else "c2", W, H, BATCH, "4:2:0" if SS else "4:4:4", wgs, label))
print("(100 MHz constant clock, 0.01 us resolution; the last eight of 64 launches issued back to back on one stream)")
names = ["wavefront runs", "kernel arguments read", "all loads issued", "first item arrived + converted", "last item converted", "barrier passed", "transform done",
         "quantised, first store next", "last store issued", "stores acknowledged"]
print("--- eight consecutive dispatches, us after the first one's first wavefront")
print("%-4s %10s %10s %12s %12s %10s %10s %10s | %s" % ("#", "first wave", "last wave", "first pixels", "last pixels", "first st.", "last st.", "last ack", "gap to previous dispatch's last ack / period"))
prev_end = prev_start = None
for k, L in enumerate(order):
    t = (raw[L] - t00) / 100.0
    st, en = t[:, :, 0].min(), t[:, :, 9].max()
    extra = "" if prev_end is None else "%+.2f / %.2f" % (st - prev_end, st - prev_start)
    print("%-4d %10.2f %10.2f %12.2f %12.2f %10.2f %10.2f %10.2f | %s" % (k, st, t[:, :, 0].max(), t[:, :, 3].min(), t[:, :, 4].max(), t[:, :, 7].min(), t[:, :, 8].max(), en, extra))
    prev_end, prev_start = en, st
L = order[-2]
t = (raw[L] - raw[L][:, :, 0].min()) / 100.0
print("--- one dispatch (the seventh), us after its first wavefront's start")
print("%-32s %8s %8s %8s %8s %8s" % ("stamp", "min", "p10", "median", "p90", "max"))
for j in range(10):
    x = t[:, :, j].ravel()
    print("%-32s %8.2f %8.2f %8.2f %8.2f %8.2f" % (names[j], x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max()))
d = t[:, :, 1] - t[:, :, 0]
print("kernel arguments: wave start -> read  min %.2f median %.2f p90 %.2f max %.2f us" % (d.min(), np.median(d), np.percentile(d, 90), d.max()))
d = t[:, :, 3] - t[:, :, 2]
print("loads issued -> first item converted  min %.2f median %.2f p90 %.2f max %.2f us" % (d.min(), np.median(d), np.percentile(d, 90), d.max()))
print("waves per microsecond bin:  not started | waiting for pixels | colour | transform | quantise | storing | done")
for b in range(0, int(t.max()) + 1):
    lo = b + 0.5
    ns = (t[:, :, 0] > lo).sum()
    wait = ((t[:, :, 0] <= lo) & (t[:, :, 3] > lo)).sum()
    col = ((t[:, :, 3] <= lo) & (t[:, :, 5] > lo)).sum()
    tr = ((t[:, :, 5] <= lo) & (t[:, :, 6] > lo)).sum()
    qu = ((t[:, :, 6] <= lo) & (t[:, :, 7] > lo)).sum()
    st = ((t[:, :, 7] <= lo) & (t[:, :, 9] > lo)).sum()
    dn = (t[:, :, 9] <= lo).sum()
    print("  t = %4.1f us   %5d %5d %5d %5d %5d %5d %5d" % (lo, ns, wait, col, tr, qu, st, dn))
# which workgroups start late?
late = np.nonzero(t[:, 0, 0] > 2.5)[0]
print("workgroups whose first wavefront starts later than 2.5 us: %d of %d; ids %s ..." % (len(late), wgs, late[:24].tolist()))
if len(late):
    print("   their ids modulo 8 (XCD): %s" % np.bincount(late % 8, minlength=8).tolist())
probe_set(None, 0)
