#!/usr/bin/env python3
"""Randomised parity stress with T CALLING THREADS at once (every thread its own context and stream; the dispatch gate between their single-pass
launches): random sizes / qualities / colour types / subsamplings / optimised tables / restart intervals, every file against the oracle.
    python tools/stress_threads.py SECONDS [THREADS] [SEED]"""
import os, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import numpy as np
import synth, oracle_lib as O
from pixo_amd import jpeg, ColorType

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
bad, counts = [], [0] * T
t0 = time.time()


def work(i):
    rng = np.random.RandomState(seed * 100 + i)
    while time.time() - t0 < budget:
        big = rng.rand() < 0.2
        w = int(rng.randint(4, 3000 if big else 500)); h = int(rng.randint(1, 1500 if big else 300))
        ct = 2 if rng.rand() < 0.8 else 0
        ss = int(rng.rand() < 0.6)
        q = int(rng.randint(1, 101)) if rng.rand() < 0.7 else int(rng.randint(90, 101))
        opt = bool(rng.rand() < 0.35)
        unit = 16 if (ss and ct == 2) else 8
        restart = None
        r = rng.rand()
        if r < 0.2: restart = int(rng.randint(1, 5)) * ((w + unit - 1) // unit)
        elif r < 0.35: restart = int(rng.randint(1, 200))
        if restart and restart > 65535: restart = None
        n = w * h * (3 if ct == 2 else 1)
        kind = int(rng.randint(0, 4))
        s = int(rng.randint(1, 1 << 30))
        if kind == 0: px = synth.lcg_bytes(n, s)
        elif kind == 1: px = (np.cumsum(synth.lcg_bytes(n, s).astype(np.int64) % 5) % 256).astype(np.uint8)
        elif kind == 2: px = ((np.arange(n, dtype=np.int64) // 3 // max(1, (s % 97) + 8)) % 256).astype(np.uint8)
        else:
            px = synth.lcg_bytes(n, s).copy(); px[px < 128] = 0; px[px >= 128] = 255
        b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss)).optimize_huffman(opt)
        if restart: b = b.restart_interval(restart)
        got = jpeg.encode(px, b.build())
        kw = dict(optimize_huffman=opt)
        if restart: kw["restart"] = restart
        want = O.encode(px, O.make_options(w, h, ct, q, ss, **kw))
        counts[i] += 1
        if got != want:
            bad.append((w, h, ct, ss, q, opt, restart, kind)); print("MISMATCH", bad[-1], flush=True)


ths = [threading.Thread(target=work, args=(i,)) for i in range(T)]
for t in ths: t.start()
for t in ths: t.join()
print("threads %d, cases %d, mismatches %d, fallbacks %d, gate (waits, timeouts) %s, %.0f s" % (T, sum(counts), len(bad), jpeg.lookback_fallbacks(), jpeg.dispatch_gate_stats(), time.time() - t0))
