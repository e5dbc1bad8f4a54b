#!/usr/bin/env python3
"""Whole baseline files from device-resident pixels into a pinned buffer (pixo_hip_jpeg_encode_device_into), 4096x4096 q=80
4:2:0: the fused pixel -> bit stream kernel (default since round 5) against the two-kernel form (debug switch
`two_kernel_scan`: coefficient kernel + scan_code), interleaved on the same box.  Content: noise (5.3 bit/px), photo
(synth.photo, ~1.3 bit/px), gradient (0.15 bit/px).
    python tools/whole_file_ab.py                 the A/B table
    python tools/whole_file_ab.py forms a= b=direct_stores ...   forms named by their PIXO_HIP_DEBUG switches
    python tools/whole_file_ab.py loop <kind> <n> [two]   n calls of one form (for rocprofv3 --kernel-trace --stats)"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import synth
from pixo_amd import jpeg

W = H = 4096
O = jpeg.JpegOptions.builder(W, H).quality(80).subsampling(jpeg.Subsampling.S420).build()


def pixels(kind):
    px = synth.noise(W, H, 42) if kind == "noise" else (synth.photo(W, H, 42) if kind == "photo" else synth.gradient_rgb(W, H))
    return torch.from_numpy(np.ascontiguousarray(px)).cuda()


def main():
    buf = torch.empty(24 << 20, dtype=torch.uint8).pin_memory()
    if len(sys.argv) > 1 and sys.argv[1] == "loop":
        kind, n = sys.argv[2], int(sys.argv[3])
        if len(sys.argv) > 4:
            jpeg.debug_configure("two_kernel_scan")
        d = pixels(kind)
        for _ in range(n):
            jpeg.encode_device_into(buf, d, O)
        print("loop", kind, n, "two_kernel_scan" if len(sys.argv) > 4 else "fused")
        return
    # forms: name=debug switches ("" = the default form); default A/B: the one-kernel form against the two-kernel form
    forms = {"fused": None, "two_kernel": "two_kernel_scan"}
    if len(sys.argv) > 1 and sys.argv[1] == "forms":
        forms = {}
        for spec in sys.argv[2:]:
            name, _, sw = spec.partition("=")
            forms[name] = sw or None
    names = list(forms)
    for kind in ("noise", "photo", "gradient"):
        d = pixels(kind)
        res = {f: [] for f in names}
        sizes = {}
        for rep in range(31):
            for form in names if rep % 2 == 0 else names[::-1]:
                jpeg.debug_configure(forms[form])
                if rep == 0:
                    for _ in range(3):
                        jpeg.encode_device_into(buf, d, O)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = jpeg.encode_device_into(buf, d, O)
                res[form].append((time.perf_counter() - t0) * 1e3)
                sig = bytes(buf[:n].numpy().tobytes()[:4096]) + n.to_bytes(8, "little")
                assert sizes.setdefault(form, sig) == sig
        jpeg.debug_configure(None)
        assert len(set(sizes.values())) == 1, "the forms give different files"
        n = int.from_bytes(sizes[names[0]][-8:], "little")
        med = {f: statistics.median(res[f][1:]) for f in names}
        print("%-9s file %9d bytes (%.2f bit/px)   " % (kind, n, n * 8 / (W * H))
              + "   ".join("%s %.4f ms (min %.4f)" % (f, med[f], min(res[f])) for f in names)
              + "   ratio %.3f" % (med[names[0]] / med[names[-1]]))


if __name__ == "__main__":
    main()
