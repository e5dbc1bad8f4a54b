"""One-off (experiment build -DPIXO_TIMELINE of jpeg_scan_fused.hip, PIXO_HIP_LIB=tools/ab/ab_timeline.so): where the time of
one baseline `scan_code` launch goes.  Every group stamps the 100 MHz constant clock at: 0 entry, 1 blocks + tables in (first
barrier), 2 walk + length scan done (second barrier), 3 bits gathered into the LDS buffer, look-back starts, 4 look-back done
(barrier), 5 bits written, 6 shared head word resolved.  Printed: per stamp the time since the launch's first stamp, as
min / median / p90 / max over groups.   python tools/scan_timeline.py [noise|gradient|flat] [size]"""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import synth
from pixo_amd import jpeg, _lib
kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
px = synth.noise(n, n, 42) if kind == "noise" else (synth.gradient_rgb(n, n) if kind == "gradient" else (synth.photo(n, n, 42) if kind == "photo" else synth.constant(n, n, 77)))
d = torch.from_numpy(px).to("cuda:0"); torch.cuda.synchronize()
prog = len(sys.argv) > 3 and sys.argv[3] == "prog"  # the progressive coder's launch (prog_code_kernel: a group per 192 blocks of a component)
b = jpeg.JpegOptions.builder(n, n).quality(80).subsampling(jpeg.Subsampling(1))
o = b.progressive(True).build() if prog else b.build()
L = ctypes.CDLL(os.environ["PIXO_HIP_LIB"])
groups = (n // 8) * (n // 8) * 3 // 2 // 192
for rep in range(6):
    f = jpeg.encode_device(d, o)
    t = np.zeros(8192 * 8, np.uint64)
    assert L.pixo_hip_debug_scan_timeline(ctypes.c_void_p(t.ctypes.data), ctypes.c_size_t(t.nbytes)) == 0
    if rep < 3: continue
    if prog:
        t = t.reshape(8192, 8)[:groups].astype(np.int64)
        t0 = t[:, 0].min()
        names = ["entry", "block in", "walks done", "run counters in", "bit counts in", "DC placed", "band 1 placed", "end"]
        print("progressive %s %dx%d, %d groups, rep %d: kernel span by stamps %.2f us" % (kind, n, n, groups, rep, (t.max() - t0) / 100.0))
        for part, sel in (("chrominance groups (two scans)", slice(0, groups // 3)), ("luminance groups (three scans)", slice(groups // 3, groups))):
            tt = t[sel]
            print("  ", part)
            for k in range(8):
                v = (tt[:, k] - t0) / 100.0
                print("   %-16s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (names[k], v.min(), np.median(v), np.percentile(v, 90), v.max()))
            dur = (tt[:, 1:] - tt[:, :-1]) / 100.0
            print("   phase medians (us):", " ".join("%s %.2f" % (names[k + 1], np.median(dur[:, k])) for k in range(7)), "| whole group %.2f" % np.median((tt[:, 7] - tt[:, 0]) / 100.0))
        continue
    t = t.reshape(8192, 8)[:groups, :7].astype(np.int64)
    t0 = t[:, 0].min()
    print("%s %dx%d, %d groups, rep %d: kernel span by stamps %.2f us" % (kind, n, n, groups, rep, (t.max() - t0) / 100.0))
    names = ["entry", "blocks in", "walk done", "gathered", "look-back done", "written", "head resolved"]
    for k in range(7):
        v = (t[:, k] - t0) / 100.0
        print("   %-15s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (names[k], v.min(), np.median(v), np.percentile(v, 90), v.max()))
    dur = (t[:, 1:] - t[:, :-1]) / 100.0
    print("   phase medians (us):", " ".join("%s %.2f" % (names[k + 1], np.median(dur[:, k])) for k in range(6)))
