"""PIXO_TIMING build probe, per wavefront: which SIMD each wave ran on, how many waves shared a
SIMD while it was in phase B, and cycles per phase (s_memtime ticks, comparable within one XCD).
Usage: PIXO_HIP_LIB=.../ab_timing.so python tools/wave_probe.py"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
NB = 2048
dbg = torch.zeros(NB * 4 * 12, dtype=torch.int64, device="cuda:0")
os.environ["PIXO_DBG_PTR"] = str(dbg.data_ptr())
import synth
from pixo_amd import jpeg
w = h = 4096
px = torch.from_numpy(synth.noise(w, h, 1)).to("cuda:0")
yb, cbn = jpeg.coefficient_geometry(w, h, 2, 1)
y = torch.empty((yb, 64), dtype=torch.int16, device="cuda:0"); cb = torch.empty((cbn, 64), dtype=torch.int16, device="cuda:0"); cr = torch.empty_like(cb)
s = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    jpeg.coefficients_device(px, w, h, 2, 1, 80, y, cb, cr, stream=s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); jpeg.coefficients_device(px, w, h, 2, 1, 80, y, cb, cr, stream=s); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
d = dbg.cpu().numpy().reshape(NB, 4, 12)
hw = d[:, :, 0]; xcc = d[:, :, 1]
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; wid = hw & 15
t0, tb, td, te = d[:, :, 2], d[:, :, 3], d[:, :, 4], d[:, :, 5]
print("kernel %.1f us" % us)
print("wave index -> SIMD histogram (rows: wave 0..3, cols: SIMD 0..3)")
for wv in range(4):
    print("  wave %d:" % wv, [int(((simd[:, wv] == k)).sum()) for k in range(4)])
print("blocks per XCC:", [int((xcc[:, 0] == k).sum()) for k in range(8)])
x0 = xcc[:, 0] == 0
span = te[x0][:, :3].max() - t0[x0].min()
print("XCC0 span %d ticks -> %.2f GHz" % (span, span / us / 1e3))
for name, a, b in (("A (start->barrier)", t0, tb), ("rows+cols", tb, td), ("quant+store", td, te), ("B total", tb, te)):
    v = (b - a)[:, :3]
    print("  %-20s mean %7.0f  p10 %7.0f  p90 %7.0f ticks" % (name, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
# concurrency on one CU of XCC 0
key = (se * 2 + sh) * 16 + cu
sel = x0
ks = np.unique(key[sel, 0])
k0 = ks[len(ks) // 2]
m = sel & (key[:, 0] == k0)
idx = np.nonzero(m)[0]
print("one CU (XCC0, key %d): %d workgroups" % (k0, len(idx)))
base = t0[idx][:, :3].min()
ev = []
for b in idx:
    for wv in range(3):
        ev.append((int(simd[b, wv]), int(tb[b, wv] - base), int(te[b, wv] - base), b, wv))
for sm in range(4):
    iv = sorted([e for e in ev if e[0] == sm], key=lambda e: e[1])
    busy = sum(e[2] - e[1] for e in iv)
    print("  SIMD %d: %d B-phase waves, sum of B durations %d ticks" % (sm, len(iv), busy))
    for e in iv[:12]:
        print("      block %4d wave %d  B from %6d to %6d (%5d)" % (e[3], e[4], e[1], e[2], e[2] - e[1]))
ws, we = d[idx][:, :3, 6], d[idx][:, :3, 7]
wall = (we.max() - ws.min()) * 10.0  # ns (100 MHz)
ticks = te[idx][:, :3].max() - t0[idx][:, :3].min()
print("  CU span: %d ticks in %.0f ns wall -> s_memtime runs at %.2f GHz" % (ticks, wall, ticks / wall))
print("  workgroup starts on this CU:", sorted(int(t0[b, 0] - base) for b in idx))
print("  per workgroup on this CU (ticks from first start): wave start / first item arrived / last item arrived / colour done / barrier passed / DCT done / end")
NW = 3
for b in sorted(idx, key=lambda b: t0[b, :NW].min()):
    for wv in range(NW):
        print("    block %4d wave %d simd %d: %6d %6d %6d %6d %6d %6d %6d" % (b, wv, simd[b, wv], t0[b, wv] - base, d[b, wv, 8] - base, d[b, wv, 9] - base,
              d[b, wv, 10] - base, tb[b, wv] - base, td[b, wv] - base, te[b, wv] - base))
# whole-GPU picture in wall-clock time (s_memrealtime, 100 MHz, common to all XCDs)
W0 = d[:, :3, 6].astype(np.int64); W1 = d[:, :3, 7].astype(np.int64)
g0 = W0.min()
print("wall clock: first wave start 0, last wave end %.2f us" % ((W1.max() - g0) / 100.0))
first_start = (W0.min(axis=1) - g0) / 100.0
last_end = (W1.max(axis=1) - g0) / 100.0
order = np.argsort(first_start)
print("  workgroup start times (us): #0 %.2f  #256 %.2f  #1024 %.2f  #1535 %.2f  #1536 %.2f  #1800 %.2f  #2047 %.2f" % tuple(first_start[order[i]] for i in (0, 256, 1024, 1535, 1536, 1800, 2047)))
es = np.sort(last_end)
print("  workgroup end times (us):   #0 %.2f  #256 %.2f  #1024 %.2f  #1535 %.2f  #1800 %.2f  #2000 %.2f  #2047 %.2f" % tuple(es[i] for i in (0, 256, 1024, 1535, 1800, 2000, 2047)))
per_cu_end = {}
kk = (xcc[:, 0] * 1024 + key[:, 0])
for k in np.unique(kk):
    per_cu_end[k] = last_end[kk == k].max()
v = np.array(sorted(per_cu_end.values()))
print("  %d CUs; CU finish time (us): min %.2f  median %.2f  p90 %.2f  max %.2f" % (len(v), v.min(), np.median(v), np.percentile(v, 90), v.max()))
cnt = np.array([int((kk == k).sum()) for k in np.unique(kk)])
print("  workgroups per CU: min %d max %d  (hist %s)" % (cnt.min(), cnt.max(), np.bincount(cnt).tolist()))
