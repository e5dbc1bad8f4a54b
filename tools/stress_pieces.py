#!/usr/bin/env python3
"""Randomised parity stress of the piece-wise delivery (not part of the test-suite): encode_device_into into pinned storage
with PIXO_HIP_DEBUG=piece_medium=2 (every scan of two or more groups in growing pieces, the coefficient kernel band by band where
MCU rows and groups share boundaries), random widths (multiples of 512 and others), heights, qualities, subsampling,
optimised tables; GPU against the oracle.  Usage: PIXO_HIP_DEBUG=piece_medium=2 stress_pieces.py SECONDS [SEED]"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import numpy as np, torch
import synth, oracle_lib as O
from pixo_amd import jpeg, ColorType

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); n = 0; bad = []
pinned = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
while time.time() - t0 < budget:
    w = int(rng.choice([512, 1024, 1536, 2048])) if rng.rand() < 0.6 else int(rng.randint(8, 2100))
    h = int(rng.randint(1, 1500))
    ct = 2 if rng.rand() < 0.8 else 0
    ss = int(rng.rand() < 0.6)
    q = int(rng.randint(1, 101))
    opt = bool(rng.rand() < 0.3)
    kind = int(rng.randint(0, 3))
    nb = w * h * (3 if ct == 2 else 1)
    px = synth.lcg_bytes(nb, int(rng.randint(1, 1 << 30)))
    if kind == 1: px = (np.cumsum(px.astype(np.int64) % 5) % 256).astype(np.uint8)
    if kind == 2: px[px < 200] = 7
    o = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss)).optimize_huffman(opt).build()
    d = torch.from_numpy(px).to("cuda:0")
    want = O.encode(px, O.make_options(w, h, ct, q, ss, optimize_huffman=opt))
    for rep in range(2):  # (the second call of a context with large files is the one cut into pieces by default)
        got_n = jpeg.encode_device_into(pinned, d, o)
        if got_n != len(want) or pinned[:got_n].numpy().tobytes() != want:
            bad.append((w, h, ct, ss, q, opt, kind, rep)); print("MISMATCH", bad[-1], flush=True)
    n += 1
print("cases %d, mismatches %d, %.0f s" % (n, len(bad), time.time() - t0))
sys.exit(1 if bad else 0)
