"""Directed coefficient tuples for the progressive coders' end-of-band run counter (a gray image's Y plane; no pixels involved):
`cases()` yields (nblocks, [(block index, kind)]) — non-empty blocks at and around wavefront (64) and group (192) boundaries,
runs of exactly 32766 / 32767 / 32768 / 65534 / 65535 empty blocks, a block that ends early right in front of such a run, the
scan's end inside a run and on a flush, random sparse patterns — and `tuple_of(nblocks, where)` builds (y, width, height)."""
import numpy as np

ZZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
      35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _block(kind, rng):
    b = np.zeros(64, np.int16)
    b[0] = rng.randint(-50, 50)
    if kind == 1:   b[ZZ[3]] = 5                      # band 1..10 only, ends early
    elif kind == 2: b[ZZ[10]] = -2; b[ZZ[63]] = 1     # both Y bands non-empty, neither ends early
    elif kind == 3: b[ZZ[40]] = 7                     # band 11..63 only, ends early
    elif kind == 4: b[ZZ[11]] = 1; b[ZZ[30]] = -300   # 16-zero runs inside the band
    return b


def tuple_of(nblocks, where, seed=4):
    rng = np.random.RandomState(seed)
    y = np.zeros((nblocks, 64), np.int16)
    y[:, 0] = rng.randint(-20, 20, nblocks)
    for i, kind in where:
        if 0 <= i < nblocks:
            y[i] = _block(kind, rng)
    cols = nblocks if nblocks <= 8000 else 4096  # a gray image of cols x rows blocks: storage order = scan order
    assert nblocks % cols == 0
    return y, 8 * cols, 8 * (nblocks // cols)


def cases(full=True):
    rng = np.random.RandomState(9)
    for edge in (63, 64, 65, 191, 192, 193, 383, 384):
        for kind in (1, 2, 3, 4):
            yield 1000, [(edge, kind)]
            yield 1000, [(edge - 1, 3), (edge, kind), (edge + 1, 1)]
    yield 192 * 3, []                                     # nothing but empty blocks: the scan's end flushes the run
    yield 192 * 3, [(192 * 3 - 1, 2)]                      # the last block is the only non-empty one
    yield 192 * 3, [(0, 1)]                                # ... the first
    for run in (32766, 32767, 32768, 65534, 65535):
        for lead in ((0, 1, 100, 191, 192) if full else (0, 191)):
            n = 4096 * 18  # 73,728 blocks
            yield n, [(lead, 1), (lead + 1 + run, 2)]      # ends early, then exactly `run` empty blocks, then a non-empty one
            yield n, [(lead, 2), (lead + 1 + run, 3), (n - 1, 1)]
    yield 4096 * 8, [(5, 2)]                               # the scan ends 32,762 empty blocks behind its only non-empty one
    yield 4096 * 8, [(0, 2)]                               # ... exactly 32,767: the flush falls on the scan's last block
    yield 4096 * 16, [(0, 1)]                              # ... 1 + 65,535: two flushes and a rest
    for density in (0.002, 0.02, 0.3):                     # random sparse patterns over many groups
        n = 4096 * 4
        idx = np.nonzero(rng.rand(n) < density)[0]
        yield n, [(int(i), int(rng.randint(1, 5))) for i in idx]
