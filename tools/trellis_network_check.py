#!/usr/bin/env python3
"""Zero-one check of the trellis kernel's 12-input sorting network (pixo_amd/csrc/jpeg_trellis.h): with the four
compare-exchanges that keep only their minimum (the maximum replaced by "largest"), wires 0..7 still carry the eight
smallest inputs in order for all 4096 zero-one inputs — hence for all inputs (min and constant are monotone)."""
import itertools

NET = [(0, 8), (1, 7), (2, 6), (3, 11), (4, 10), (5, 9), (0, 1), (2, 5), (3, 4), (6, 9), (7, 8), (10, 11), (0, 2), (1, 6), (5, 10),
       (9, 11), (0, 3), (1, 2), (4, 6), (5, 7), (8, 11), (9, 10), (1, 4), (3, 5), (6, 8), (7, 10), (1, 3), (2, 5), (6, 9), (8, 10),
       (2, 3), (4, 5), (6, 7), (8, 9), (4, 6), (5, 7), (3, 4), (5, 6), (7, 8)]
MIN_ONLY = {20, 29, 33, 38}  # (8, 11), (8, 10), (8, 9), (7, 8)

bad = 0
for v in itertools.product([0, 1], repeat=12):
    w = list(v)
    for i, (a, b) in enumerate(NET):
        lo, hi = min(w[a], w[b]), max(w[a], w[b])
        w[a], w[b] = lo, (1 if i in MIN_ONLY else hi)
    bad += w[:8] != sorted(v)[:8]
print("inputs with a wrong first eight:", bad)
assert bad == 0
