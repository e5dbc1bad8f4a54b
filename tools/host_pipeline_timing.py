#!/usr/bin/env python3
"""encode() / encode_into() from HOST pixels: the banded upload pipeline (pixels in over PCIe band by band, each band
transformed and coded while the next one travels, coded pieces on their way back meanwhile) against one upload copy
(PIXO_HIP_DEBUG=no_bands_upload).  4096x4096 and 16384x16384 noise + a smooth image; bytes checked against each other."""
import os, sys, time, hashlib
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import synth
from pixo_amd import jpeg


def med(fn, reps):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2], min(ts), r


for size, kind, reps in ((2048, "noise", 9), (4096, "noise", 9), (4096, "smooth", 9), (16384, "noise", 3)):
    px = synth.noise(size, size, 42) if kind == "noise" else synth.gradient_rgb(size, size)
    o = jpeg.JpegOptions.builder(size, size).quality(80).subsampling(jpeg.Subsampling.S420).build()
    pin = torch.empty(size * size * 2, dtype=torch.uint8).pin_memory()
    pageable = np.empty(size * size * 2, np.uint8)
    digests = set()
    for mode in ("", "no_bands_upload"):
        jpeg.debug_configure(mode)
        a = med(lambda: jpeg.encode(px, o), reps)
        b = med(lambda: jpeg.encode_into_buffer(pin.numpy(), px, o), reps)
        c = med(lambda: jpeg.encode_into_buffer(pageable, px, o), reps)
        digests.add(hashlib.sha256(a[2]).hexdigest()); digests.add(hashlib.sha256(pin.numpy()[:b[2]].tobytes()).hexdigest())
        digests.add(hashlib.sha256(pageable[:c[2]].tobytes()).hexdigest())
        print("%5d^2 %-6s %-16s encode() %8.3f ms (min %8.3f) | encode_into pinned %8.3f (min %8.3f) | encode_into pageable %8.3f (min %8.3f) | %d bytes"
              % (size, kind, mode or "banded upload", a[0], a[1], b[0], b[1], c[0], c[1], len(a[2])), flush=True)
    assert len(digests) == 1, digests
    jpeg.debug_configure(None)
    del px, pin, pageable
