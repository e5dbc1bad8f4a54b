#!/usr/bin/env python3
"""Randomised parity stress (not part of the test-suite: minutes of oracle time): JPEG whole files with
random shapes / qualities / flags and PNG filter streams with random shapes / strategies / pixel sizes,
GPU against the oracle.  Usage: stress_parity.py SECONDS [SEED] [progressive]   (progressive: every JPEG case is a progressive
file, sizes and contents chosen so that the scans span many groups of the single-pass coder — long runs of empty blocks, a few
symbols here and there; no PNG cases).  Also fails when a single-pass entropy launch fell back to the multi-pass kernels."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import numpy as np
import synth, oracle_lib as O
from pixo_amd import jpeg, png, ColorType

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
prog_only = len(sys.argv) > 3 and sys.argv[3] == "progressive"
fused_only = len(sys.argv) > 3 and sys.argv[3] == "fused"  # only what the fused pixel -> scan kernel serves: RGB, baseline, standard tables
t0 = time.time(); nj = npn = 0; bad = []

def content(n, kind, seed):
    if kind == 0: return synth.lcg_bytes(n, seed)
    if kind == 1: return (np.cumsum(synth.lcg_bytes(n, seed).astype(np.int64) % 5) % 256).astype(np.uint8)
    if kind == 2: return np.full(n, seed % 256, np.uint8)
    if kind == 4:  # flat with sparse specks: most blocks empty in every band, a symbol now and then
        b = np.full(n, seed % 256, np.uint8); k = synth.lcg_bytes(n, seed); b[k < 2] = 255 - (seed % 256); return b
    if kind == 5:  # a slow ramp: only DC changes, end-of-band runs of thousands of blocks
        return ((np.arange(n, dtype=np.int64) // 3 // max(1, (seed % 97) + 8)) % 256).astype(np.uint8)
    b = synth.lcg_bytes(n, seed); b[b < 128] = 0; b[b >= 128] = 255; return b

while time.time() - t0 < budget:
    # ---- JPEG
    big = rng.rand() < (0.6 if prog_only else 0.15)
    w = int(rng.randint(1, 2600 if big else 400)); h = int(rng.randint(1, 1400 if big else 300))
    ct = 2 if rng.rand() < 0.75 else 0
    ss = int(rng.rand() < 0.6)
    q = int(rng.randint(1, 101))
    flags = dict(optimize_huffman=bool(rng.rand() < 0.4), progressive=bool(prog_only or rng.rand() < 0.3), trellis=bool(rng.rand() < 0.3))
    restart = int(rng.randint(1, 40)) if rng.rand() < 0.25 else None
    if fused_only:
        ct, restart = 2, None
        flags = dict(optimize_huffman=False, progressive=False, trellis=False)
        big = rng.rand() < 0.5
        w = int(rng.randint(4, 4200 if big else 600)); h = int(rng.randint(1, 1200 if big else 300))
        if rng.rand() < 0.3: q = int(rng.randint(90, 101))  # long blocks, groups of several rounds, many 0xFF bytes
    if fused_only and rng.rand() < 0.35:  # restart intervals of whole MCU rows: segments of the fused kernel (round 6)
        unit = 16 if ss else 8
        restart = int(rng.randint(1, 6)) * ((w + unit - 1) // unit)
        if restart > 65535: restart = None
    kind = int(rng.randint(0, 8))
    if ct == 2 and kind >= 6:  # photograph-like content (synth.scene / synth.photo: edges, texture, saturated details)
        px = synth.scene(w, h, int(rng.randint(1, 1 << 20))) if kind == 6 else synth.photo(w, h, int(rng.randint(1, 1 << 20)))
    else:
        px = content(w * h * (3 if ct == 2 else 1), kind % 6, int(rng.randint(1, 1 << 30)))
    b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss)) \
        .optimize_huffman(flags["optimize_huffman"]).progressive(flags["progressive"]).trellis_quant(flags["trellis"])
    if restart: b = b.restart_interval(restart)
    got = jpeg.encode(px, b.build())
    kw = dict(flags); 
    if restart: kw["restart"] = restart
    want = O.encode(px, O.make_options(w, h, ct, q, ss, **kw))
    nj += 1
    if got != want:
        bad.append(("jpeg", w, h, ct, ss, q, flags, restart)); print("MISMATCH", bad[-1], flush=True)
    if prog_only or fused_only:
        continue
    # ---- PNG
    bpp = int(rng.choice([1, 2, 3, 4, 6, 8])); w = int(rng.randint(1, 6000 if rng.rand() < 0.2 else 700)); h = int(rng.randint(1, 120))
    st = int(rng.randint(0, 9))
    px = content(w * h * bpp, int(rng.randint(0, 4)), int(rng.randint(1, 1 << 30)))
    want, wad = O.png_filter(px, w, h, bpp, st, stateful_fast=(h <= 32))
    got, gad = png.apply_filters(px, w, h, bpp, st)
    npn += 1
    if not (np.array_equal(got, want) and gad == wad):
        bad.append(("png", w, h, bpp, st)); print("MISMATCH", bad[-1], flush=True)
fb = jpeg.lookback_fallbacks()
print("jpeg cases %d, png cases %d, mismatches %d, single-pass launches that fell back to the multi-pass kernels %d, %.0f s"
      % (nj, npn, len(bad), fb, time.time() - t0))
sys.exit(1 if bad or fb else 0)
