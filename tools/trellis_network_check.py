#!/usr/bin/env python3
"""Zero-one check of the trellis kernel's 11-input sorting network (pixo_amd/csrc/jpeg_trellis.h, round 4: the round and
the ceil candidate exclude each other, so eleven of the twelve slots are ever filled): with the three compare-exchanges
that keep only their minimum (the maximum replaced by "largest"), wires 0..7 still carry the eight smallest inputs in order
for all 2048 zero-one inputs — hence for all inputs (min and constant are monotone)."""
import itertools

NET = [(0, 9), (1, 6), (2, 4), (3, 7), (5, 8), (0, 1), (3, 5), (4, 10), (6, 9), (7, 8), (1, 3), (2, 5), (4, 7), (8, 10),
       (0, 4), (1, 2), (3, 7), (5, 9), (6, 8), (0, 1), (2, 6), (4, 5), (7, 8), (9, 10), (2, 4), (3, 6), (5, 7), (8, 9),
       (1, 2), (3, 4), (5, 6), (7, 8), (2, 3), (4, 5), (6, 7)]
MIN_ONLY = {23, 27, 31}  # (9, 10), (8, 9), the last (7, 8)

bad = 0
for v in itertools.product([0, 1], repeat=11):
    w = list(v)
    for i, (a, b) in enumerate(NET):
        lo, hi = min(w[a], w[b]), max(w[a], w[b])
        w[a], w[b] = lo, (1 if i in MIN_ONLY else hi)
    bad += w[:8] != sorted(v)[:8]
print("inputs with a wrong first eight:", bad)
assert bad == 0
