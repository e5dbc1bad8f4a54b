"""One-off: progressive files of very large images — the single-pass coder against the library's host twin (jpeg_host.cpp,
PIXO_HIP_DEBUG=host_entropy), whole-file bytes.  16384x16384: 1,048,576 luminance blocks per scan = 5,462 groups, end-of-band
runs far beyond 32767 on smooth content.   python tools/prog_large_check.py [size]"""
import hashlib, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import synth
from pixo_amd import jpeg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for kind in ("noise", "gradient", "flat"):
    px = synth.noise(n, n, 42) if kind == "noise" else (synth.gradient_rgb(n, n) if kind == "gradient" else synth.constant(n, n, 77))
    d = torch.from_numpy(px).to("cuda:0"); torch.cuda.synchronize()
    for ss in (1, 0):
        o = jpeg.JpegOptions.builder(n, n).quality(80).subsampling(jpeg.Subsampling(ss)).progressive(True).build()
        jpeg.debug_configure("")
        t0 = time.perf_counter(); dev = jpeg.encode_device(d, o); t_dev = time.perf_counter() - t0
        jpeg.debug_configure("host_entropy")
        t0 = time.perf_counter(); host = jpeg.encode_device(d, o); t_host = time.perf_counter() - t0
        jpeg.debug_configure("")
        same = dev == host
        print("%dx%d %-8s %s: %d bytes, device %.1f ms, host twin %.1f ms, identical %s, fallbacks %d" %
              (n, n, kind, "4:2:0" if ss else "4:4:4", len(dev), t_dev * 1e3, t_host * 1e3, same, jpeg.lookback_fallbacks()), flush=True)
        assert same
    del d
    torch.cuda.empty_cache()
