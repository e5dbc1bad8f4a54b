#!/usr/bin/env python3
"""pixo_hip_jpeg_encode_multi on ONE GPU with 1, 2, 4, 8 bands (each band its own thread, context and stream) against
pixo_hip_jpeg_encode: 4096x4096 and 16384x16384 noise from host pixels.  With PIXO_HIP_DEBUG=trace the per-phase times."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from pixo_amd import jpeg

for size in (4096, 16384):
    px = synth.noise(size, size, 42)
    o = jpeg.JpegOptions.builder(size, size).quality(80).subsampling(jpeg.Subsampling.S420).build()
    want = jpeg.encode(px, o)
    reps = 5 if size == 4096 else 2

    def t(fn):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); blob = fn(); ts.append(time.perf_counter() - t0)
        assert blob == want
        return min(ts) * 1e3

    print("%5d^2  encode()            %8.2f ms" % (size, t(lambda: jpeg.encode(px, o))), flush=True)
    for parts in (1, 2, 4, 8):
        print("%5d^2  encode_multi x%d     %8.2f ms" % (size, parts, t(lambda: jpeg.encode_multi(px, o, [0] * parts))), flush=True)
