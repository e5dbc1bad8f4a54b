#!/usr/bin/env python3
"""One 4096x4096 image as ONE launch of the coefficient kernel against the same image as TWO half-image launches (MCU rows
are independent) on one stream, on two streams, and on two streams of different priority: does offsetting the halves'
read and write phases buy anything?    Wall clock over many images at steady clocks."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from pixo_amd import jpeg
W = H = 4096
dev = torch.device("cuda", 0)
yb, cbn = jpeg.coefficient_geometry(W, H, 2, 1)
base = torch.from_numpy(np.ascontiguousarray(synth.noise(W, H, 42))).to(dev)
NB = 7
ins = [(base ^ torch.tensor(i, dtype=torch.uint8, device=dev)).contiguous() for i in range(NB)]
outs = [tuple(torch.empty((n, 64), dtype=torch.int16, device=dev) for n in (yb, cbn, cbn)) for _ in range(NB)]
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
s_main = torch.cuda.Stream()
s_a, s_b = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
s_c = torch.cuda.Stream(priority=0)


def whole(i, stream):
    y, cb, cr = outs[i % NB]
    jpeg.coefficients_device(ins[i % NB], W, H, 2, 1, 80, y, cb, cr, stream=stream.cuda_stream)


def part(i, stream, k, parts):
    y, cb, cr = outs[i % NB]
    rows = H // parts
    px = ins[i % NB].data_ptr() + k * rows * W * 3
    mcus = (rows // 16) * (W // 16) * k
    jpeg.coefficients_device(px, W, rows, 2, 1, 80, y.data_ptr() + mcus * 4 * 128, cb.data_ptr() + mcus * 128, cr.data_ptr() + mcus * 128, stream=stream.cuda_stream)


def run(name, step, n=600):
    for i in range(200):
        step(i)
    torch.cuda.synchronize()
    best = []
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(n):
            step(i)
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / n * 1e6)
    print("%-58s %6.2f us per image (median of 5: min %.2f max %.2f)" % (name, sorted(best)[2], min(best), max(best)), flush=True)


run("one launch, one stream", lambda i: whole(i, s_main))
run("two half-image launches, one stream", lambda i: (part(i, s_main, 0, 2), part(i, s_main, 1, 2)))
run("two half-image launches, two streams (equal priority)", lambda i: (part(i, s_b, 0, 2), part(i, s_c, 1, 2)))
run("two half-image launches, two streams (high / normal priority)", lambda i: (part(i, s_a, 0, 2), part(i, s_b, 1, 2)))
run("four quarter launches, two streams alternating", lambda i: (part(i, s_b, 0, 4), part(i, s_c, 1, 4), part(i, s_b, 2, 4), part(i, s_c, 3, 4)))
run("whole images alternating over two streams", lambda i: whole(i, s_b if i & 1 else s_c))
run("one launch, one stream (again)", lambda i: whole(i, s_main))
