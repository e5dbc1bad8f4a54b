"""One pixo_hip_jpeg_encode_multi call of the 16384x16384 image on device 0 with `parts` bands, wall times of its phases
(band encoder steps timed from Python through the same C ABI the library's own band threads use)."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth
from pixo_amd import jpeg
size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
px = synth.noise(size, size, 42)
o = jpeg.JpegOptions.builder(size, size).quality(80).subsampling(jpeg.Subsampling.S420).build()
for parts in (1, 8):
    jpeg.encode_multi(px, o, [0] * parts)
    t0 = time.perf_counter(); jpeg.encode_multi(px, o, [0] * parts); print("encode_multi x%d: %.2f ms" % (parts, (time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); jpeg.encode(px, o); jpeg.encode(px, o); print("encode(): %.2f ms per call" % ((time.perf_counter() - t0) * 500))
# the steps of ONE band = the whole image, by hand
enc = jpeg.BandEncoder(o, 1, 0, 0)
for rep in range(2):
    t = [time.perf_counter()]
    last = enc.coeffs(px); t.append(time.perf_counter())
    bits = enc.lengths([0, 0, 0]); t.append(time.perf_counter())
    hdr, n = enc.pack_device(0); t.append(time.perf_counter())
    out = np.empty(n + 64, np.uint8) if rep == 0 else out
    enc.copy_body(out); t.append(time.perf_counter())
    print("band encoder, one band: coeffs (upload + kernel) %.2f  lengths %.2f  pack %.2f  copy_body (%d MB, %s pages) %.2f ms"
          % tuple([(t[i + 1] - t[i]) * 1e3 for i in range(3)] + [n >> 20, "fresh" if rep == 0 else "touched", (t[4] - t[3]) * 1e3]))
enc.close()
