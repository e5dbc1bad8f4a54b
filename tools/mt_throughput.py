#!/usr/bin/env python3
"""Whole files from HOST pixels, T threads at once (how the reference's rayon users call encode):
every thread has its own context (stream, device and pinned buffers) inside the library and calls
pixo_hip_jpeg_encode_into with a buffer it reuses.  Prints files/s and Mpixels/s per thread count."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import synth  # noqa: E402
from pixo_amd import _lib, jpeg  # noqa: E402

W = H = int(os.environ.get("SIZE", "4096"))
KIND = os.environ.get("KIND", "noise")
REPS = int(os.environ.get("REPS", "12"))
L = _lib.load()
px = synth.noise(W, H, 42) if KIND == "noise" else synth.gradient_rgb(W, H)
opts = jpeg.JpegOptions.builder(W, H).quality(80).subsampling(jpeg.Subsampling.S420).build()
co = opts._c() if hasattr(opts, "_c") else None


def c_options(o):
    s = _lib.JpegOptionsC()
    L.pixo_jpeg_options_from_preset(C.byref(s), o.width, o.height, o.quality, 0)
    s.subsampling = int(o.subsampling)
    s.color_type = int(o.color_type)
    return s


def worker(n, out_sizes, start, idx):
    o = c_options(opts)
    out = np.empty(W * H * 3 // 2 + 4096, np.uint8)
    need = C.c_size_t()
    mine = px.copy()  # each thread its own source pages
    rc = L.pixo_hip_jpeg_encode_into(out.ctypes.data, out.size, mine.ctypes.data, mine.size, C.byref(o), C.byref(need))
    assert rc == 0, rc
    start.wait()
    for _ in range(n):
        rc = L.pixo_hip_jpeg_encode_into(out.ctypes.data, out.size, mine.ctypes.data, mine.size, C.byref(o), C.byref(need))
        assert rc == 0, rc
    out_sizes[idx] = need.value


L.pixo_hip_jpeg_encode_into.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
for T in (1, 2, 4, 8, 16):
    sizes = [0] * T
    start = threading.Barrier(T + 1)
    th = [threading.Thread(target=worker, args=(REPS, sizes, start, i)) for i in range(T)]
    for t in th: t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    files = T * REPS
    print("%2d threads: %7.1f files/s  %9.1f Mpixels/s  (%.2f ms per file per thread, %d bytes)" % (
        T, files / dt, files * W * H / dt / 1e6, dt / REPS * 1e3, sizes[0]), flush=True)
