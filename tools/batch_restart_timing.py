#!/usr/bin/env python3
"""Whole-file timings of the segmented single-pass entropy path: 64 x 1080p batches (malloc'd files / one pinned arena), restart
intervals at 4096x4096 against the plain file, each also with the multi-pass kernels (PIXO_HIP_DEBUG=multipass_entropy)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from pixo_amd import jpeg
dev = torch.device("cuda", 0)


def med(f, n=9):
    f(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[n // 2], min(ts)


def batch_case(kind):
    w, h, n = 1920, 1080, 64
    base = synth.noise(w, h, 42) if kind == "noise" else synth.gradient_rgb(w, h)
    d = torch.cat([torch.from_numpy(base).to(dev) ^ torch.tensor(i if kind == "noise" else 0, dtype=torch.uint8, device=dev) for i in range(n)]).contiguous()
    o = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420).build()
    arena = torch.empty(n * w * h, dtype=torch.uint8).pin_memory()
    for mode in ("", "multipass_entropy"):
        jpeg.debug_configure(mode)
        a = med(lambda: jpeg.encode_batch_device_into(arena, d, o, n))
        import ctypes as C
        from pixo_amd import _lib
        L = _lib.load(); files = (C.POINTER(C.c_uint8) * n)(); lens_c = (C.c_size_t * n)(); oc = o._c()
        def c_call():
            rc = L.pixo_hip_jpeg_encode_batch_device(d.data_ptr(), C.byref(oc), n, files, lens_c)
            assert rc == 0
            for i in range(n): L.pixo_hip_free(files[i])
        b = med(c_call, 5)
        offs, lens = jpeg.encode_batch_device_into(arena, d, o, n)
        pageable = np.empty(sum(lens) + 64, np.uint8)  # (a caller's own, resident after the first call)
        p = med(lambda: jpeg.encode_batch_device_into(pageable, d, o, n), 5)
        print("batch 64 x 1080p %-8s %-18s into pinned arena %7.3f ms (min %7.3f)   into a pageable arena %7.3f ms (min %7.3f)   64 malloc'd files (C call) %7.3f ms (min %7.3f)   %d bytes"
              % (kind, mode or "single-pass", a[0], a[1], p[0], p[1], b[0], b[1], sum(lens)), flush=True)
    jpeg.debug_configure(None)


def restart_case():
    w = h = 4096
    d = torch.from_numpy(synth.noise(w, h, 42)).to(dev)
    pin = torch.empty(w * h, dtype=torch.uint8).pin_memory()
    for mode in ("", "multipass_entropy"):
        jpeg.debug_configure(mode)
        for r in (None, 256, 64, 16, 8):
            b = jpeg.JpegOptions.builder(w, h).quality(80).subsampling(jpeg.Subsampling.S420)
            if r: b = b.restart_interval(r)
            o = b.build()
            t = med(lambda: jpeg.encode_device_into(pin, d, o))
            print("4096x4096 noise restart %-5s %-18s device pixels -> pinned file %7.3f ms (min %7.3f)" % (r, mode or "single-pass", t[0], t[1]), flush=True)
    jpeg.debug_configure(None)


batch_case("noise"); batch_case("smooth"); restart_case()
