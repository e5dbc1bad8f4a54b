"""Where the time of one `pixels_code_kernel` launch goes (experiment build: AB_SRC=jpeg_pixels_code.hip tools/ab_build.sh
timeline -DPIXO_TIMELINE; PIXO_HIP_LIB=tools/ab/ab_timeline.so).  Thread 0 of every group stamps the 100 MHz constant clock at:
0 entry, 1 phase A done (pixels in LDS), 2 phase B done (quantised), 3 walk done, 4 scan-order prefix done, 5 window gathered,
6 first look-back, 7 census, 8 second look-back, 9 expanded, 10 stored
(PIXO_TIMELINE_R05=1: the round-5 kernel's stamps — 6 first look-back, 7 census, 8 second look-back, 9 stored).  Printed per stamp: time since the
launch's first stamp as min / median / p90 / max over groups, and the median duration of every phase.
    python tools/pixels_code_timeline.py [noise|photo|gradient] [size]"""
import ctypes
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import torch
import synth
from pixo_amd import jpeg

kinds = sys.argv[1].split(",") if len(sys.argv) > 1 else ["noise", "photo", "gradient"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
L = ctypes.CDLL(os.environ["PIXO_HIP_LIB"])
names = ["entry", "phase A", "phase B", "walk", "prefix", "gathered", "look-back 1", "census", "look-back 2", "expanded", "stored"] if not os.environ.get("PIXO_TIMELINE_R05") else ["entry", "phase A", "phase B", "walk", "prefix", "gathered", "look-back 1", "census", "look-back 2", "stored"]
NS = len(names)
o = jpeg.JpegOptions.builder(n, n).quality(int(os.environ.get("Q", "80"))).subsampling(jpeg.Subsampling(1)).build()
groups = ((n + 511) // 512) * ((n + 15) // 16)
stream = torch.cuda.current_stream().cuda_stream
for kind in kinds:
    px = synth.noise(n, n, 42) if kind == "noise" else (synth.gradient_rgb(n, n) if kind == "gradient" else synth.photo(n, n, 42))
    d = torch.from_numpy(np.ascontiguousarray(px)).to("cuda:0")
    torch.cuda.synchronize()
    for rep in range(8):
        jpeg.debug_scan_device_async(d, o, stream=stream)
        torch.cuda.synchronize()
        t = np.zeros(8192 * 16, np.uint64)
        assert L.pixo_hip_debug_pixels_code_timeline(ctypes.c_void_p(t.ctypes.data), ctypes.c_size_t(t.nbytes)) == 0
        if rep < 6:
            continue
        t = t.reshape(8192, 16)[:groups, :NS].astype(np.int64)
        t0 = t[:, 0].min()
        print("%s %dx%d, %d groups, rep %d: kernel span by stamps %.2f us" % (kind, n, n, groups, rep, (t.max() - t0) / 100.0))
        for k in range(NS):
            v = (t[:, k] - t0) / 100.0
            print("   %-12s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (names[k], v.min(), np.median(v), np.percentile(v, 90), v.max()))
        dur = (t[:, 1:] - t[:, :-1]) / 100.0
        print("   phase medians (us):", " ".join("%s %.2f" % (names[k + 1], np.median(dur[:, k])) for k in range(NS - 1)))
        # by dispatch order: the first / middle / last 256 groups
        for label, sel in (("groups 0..255", slice(0, 256)), ("groups 896..1151", slice(896, 1152)), ("last 256 groups", slice(groups - 256, groups))):
            tt = t[sel]
            print("   %-16s" % label, " ".join("%s %.1f" % (names[k], np.median((tt[:, k] - t0) / 100.0)) for k in range(NS)))
    del d
