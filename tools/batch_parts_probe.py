#!/usr/bin/env python3
"""configs[2] as whole files (64 x 1920x1080 device-resident images -> one pinned arena, pixo_hip_jpeg_encode_batch_device_into)
with the batch cut into 1 / 2 / 4 / 8 sub-batches (debug switch batch_parts=N: a sub-batch's copy runs while the next one's
kernels do), per content — where the library's rule (8 sub-batches above 8 bytes per block, else one) should switch.
    python tools/batch_parts_probe.py [parts ...]"""
import hashlib
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

W, H, N = 1920, 1080, 64
O = jpeg.JpegOptions.builder(W, H).quality(80).subsampling(jpeg.Subsampling.S420).build()


def main():
    parts = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8]
    arena = torch.empty(N * W * H, dtype=torch.uint8).pin_memory()
    for kind in ("noise", "photo", "gradient"):
        px = synth.noise(W, H, 42) if kind == "noise" else (synth.photo(W, H, 42) if kind == "photo" else synth.gradient_rgb(W, H))
        d = torch.from_numpy(np.ascontiguousarray(px)).cuda().repeat(N).contiguous()
        res = {p: [] for p in parts}
        sig = {}
        for rep in range(16):
            for p in parts if rep % 2 == 0 else parts[::-1]:
                jpeg.debug_configure("batch_parts=%d" % p if p else None)
                if rep == 0:
                    jpeg.encode_batch_device_into(arena, d, O, N)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                offs, lens = jpeg.encode_batch_device_into(arena, d, O, N)
                res[p].append((time.perf_counter() - t0) * 1e3)
                if rep == 1:
                    sig[p] = hashlib.sha256(arena[: offs[-1] + lens[-1]].numpy().tobytes()).hexdigest()
        jpeg.debug_configure(None)
        assert len(set(sig.values())) == 1, "the part counts give different arenas"
        total = int(sum(lens))
        print("%-9s %9d bytes (%.1f bytes per block)  " % (kind, total, total / (N * (W // 8) * ((H + 7) // 8) * 1.5))
              + "   ".join("%s %.3f ms (min %.3f)" % ("library" if p == 0 else "parts=%d" % p, statistics.median(res[p][1:]), min(res[p])) for p in parts))
        del d
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
