#!/usr/bin/env python3
"""Randomised parity stress of BATCHES (pixo_hip_jpeg_encode_batch_device / _into: N equal images in HBM -> N files; by default one launch of the fused
kernel with every image a segment, or the two-kernel form for narrow images) and of the device-pointer single-image entries: random sizes, counts,
qualities, colour types, subsamplings; every file against the oracle.   python tools/stress_batches.py SECONDS [SEED]"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import numpy as np, torch
import synth, oracle_lib as O
from pixo_amd import jpeg, ColorType

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); nb = nf = 0; bad = []
while time.time() - t0 < budget:
    big = rng.rand() < 0.25
    w = int(rng.randint(4, 2200 if big else 700)); h = int(rng.randint(1, 1200 if big else 300))
    n = int(rng.randint(2, 6 if big else 40))
    ct = 2 if rng.rand() < 0.8 else 0
    ss = int(rng.rand() < 0.6)
    q = int(rng.randint(1, 101)) if rng.rand() < 0.75 else int(rng.randint(92, 101))
    per = w * h * (3 if ct == 2 else 1)
    kind = int(rng.randint(0, 3))
    imgs = []
    for i in range(n):
        s = int(rng.randint(1, 1 << 30))
        if kind == 0: px = synth.lcg_bytes(per, s)
        elif kind == 1: px = (np.cumsum(synth.lcg_bytes(per, s).astype(np.int64) % 5) % 256).astype(np.uint8)
        else: px = ((np.arange(per, dtype=np.int64) // 3 // max(1, (s % 97) + 8)) % 256).astype(np.uint8)
        imgs.append(px)
    d = torch.from_numpy(np.concatenate(imgs)).cuda()
    o = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss)).build()
    want = [O.encode(px, O.make_options(w, h, ct, q, ss)) for px in imgs]
    form = int(rng.randint(0, 3))
    if form == 0:
        got = [bytes(f) for f in jpeg.encode_batch_device(d, o, n)]
    elif form == 1:
        arena = torch.empty(sum(len(f) for f in want) + 64, dtype=torch.uint8).pin_memory()
        offs, lens = jpeg.encode_batch_device_into(arena, d, o, n)
        a = arena.numpy(); got = [a[offs[i]: offs[i] + lens[i]].tobytes() for i in range(n)]
    else:
        arena = torch.empty(sum(len(f) for f in want) + 64, dtype=torch.uint8, device="cuda")
        offs, lens = jpeg.encode_batch_device_into(arena, d, o, n)
        a = arena.cpu().numpy(); got = [a[offs[i]: offs[i] + lens[i]].tobytes() for i in range(n)]
    nb += 1; nf += n
    if got != want:
        bad.append((w, h, n, ct, ss, q, kind, form)); print("MISMATCH batch", bad[-1], flush=True)
    # the first image alone through the device-pointer entries, with optimised tables now and then
    opt = bool(rng.rand() < 0.4)
    o1 = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss)).optimize_huffman(opt).build()
    d1 = d[:per]
    w1 = want[0] if not opt else O.encode(imgs[0], O.make_options(w, h, ct, q, ss, optimize_huffman=True))
    if jpeg.encode_device(d1, o1) != w1:
        bad.append(("single", w, h, ct, ss, q, opt)); print("MISMATCH single", bad[-1], flush=True)
    del d
print("batches %d, files %d, mismatches %d, fallbacks %d, %.0f s" % (nb, nf, len(bad), jpeg.lookback_fallbacks(), time.time() - t0))
