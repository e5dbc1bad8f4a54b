"""Deterministic synthetic inputs, integer-only, no RNG library.

Same generators the reference's own tests and benches use, restated in numpy:
  noise            -> tests/support/synthetic.rs:183  (LCG, one step per byte)
  noise_gray       -> tests/support/synthetic.rs:200
  gradient_rgb     -> benches/comparison.rs:32 (generate_gradient_image)
  flat_blocks      -> benches/comparison.rs:82 (generate_flat_blocks_image)
  checkerboard     -> tests/support/synthetic.rs:88
  rgba_noise_alpha1-> SURVEY.md §8c C5 input (LCG bytes, every 4th byte |= 1)
"""
import numpy as np

_A = 1103515245
_C = 12345


def lcg_bytes(n: int, seed: int) -> np.ndarray:
    """n bytes of `state = state*1103515245 + 12345 (mod 2^32); byte = state>>16`.

    Vectorised with the closed form state_k = A^k*seed + C*(A^k-1)/(A-1) computed
    by blocked affine-map composition (all arithmetic mod 2^32 in uint64 lanes).
    """
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    out = np.empty(n, dtype=np.uint8)
    # affine maps x -> a*x + c for 1..B steps
    B = 1 << 16
    a = np.empty(B, dtype=np.uint64)
    c = np.empty(B, dtype=np.uint64)
    ca, cc = 1, 0
    M = 0xFFFFFFFF
    # build tables by doubling: (a,c) for k steps
    a_list = [1]
    c_list = [0]
    # sequentially is O(B) python ops = 65k, fine
    for k in range(B):
        ca = (ca * _A) & M
        cc = (cc * _A + _C) & M
        a[k] = ca
        c[k] = cc
    state = seed & M
    pos = 0
    while pos < n:
        m = min(B, n - pos)
        s = (a[:m] * np.uint64(state) + c[:m]) & np.uint64(M)
        out[pos:pos + m] = ((s >> np.uint64(16)) & np.uint64(0xFF)).astype(np.uint8)
        state = int(s[m - 1])
        pos += m
    return out


def lcg_skip(seed: int, k: int) -> int:
    """The generator's state after k steps (affine map composed by repeated squaring, mod 2^32)."""
    M = 0xFFFFFFFF
    a, c = _A, _C          # one step: x -> a x + c
    ra, rc = 1, 0          # identity
    while k:
        if k & 1:
            ra, rc = (a * ra) & M, (a * rc + c) & M
        a, c = (a * a) & M, (a * c + c) & M
        k >>= 1
    return (ra * (seed & M) + rc) & M


def noise(w: int, h: int, seed: int = 42) -> np.ndarray:
    return lcg_bytes(w * h * 3, seed)


def noise_rows(w: int, h: int, seed: int, row_begin: int, row_end: int) -> np.ndarray:
    """Rows [row_begin, row_end) of noise(w, h, seed) without generating the rows above them."""
    return lcg_bytes((row_end - row_begin) * w * 3, lcg_skip(seed, row_begin * w * 3))


def extremes(w: int, h: int, seed: int = 42) -> np.ndarray:
    """RGB noise whose channels only take the values 0, 1, 254, 255: every pixel sits on or next
    to a corner of the colour cube, so neighbouring pixels mix clamped and unclamped chroma."""
    return np.array([0, 1, 254, 255], np.uint8)[lcg_bytes(w * h * 3, seed) >> 6]


def noise_gray(w: int, h: int, seed: int = 42) -> np.ndarray:
    return lcg_bytes(w * h, seed)


def rgba_noise_alpha1(w: int, h: int, seed: int = 42) -> np.ndarray:
    b = lcg_bytes(w * h * 4, seed)
    b[3::4] |= 1
    return b


def gradient_rgb(w: int, h: int) -> np.ndarray:
    x = np.arange(w, dtype=np.uint32)[None, :]
    y = np.arange(h, dtype=np.uint32)[:, None]
    r = ((x * 255) // max(w, 1)).astype(np.uint8) + np.zeros((h, 1), np.uint8)
    g = ((y * 255) // max(h, 1)).astype(np.uint8) + np.zeros((1, w), np.uint8)
    b = (((x + y) * 127) // max(w + h, 1)).astype(np.uint8)
    return np.stack([r, g, b], axis=-1).reshape(-1)


def flat_blocks(w: int, h: int) -> np.ndarray:
    colors = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0]], np.uint8)
    bx = (np.arange(w) >= w // 2).astype(np.int64)[None, :]
    by = (np.arange(h) >= h // 2).astype(np.int64)[:, None]
    return colors[by * 2 + bx].reshape(-1)


def checkerboard(w: int, h: int, cell: int = 8) -> np.ndarray:
    cell = max(cell, 1)
    cx = (np.arange(w) // cell)[None, :]
    cy = (np.arange(h) // cell)[:, None]
    v = np.where((cx + cy) % 2 == 0, 255, 0).astype(np.uint8)
    return np.repeat(v.reshape(-1), 3)


def constant(w: int, h: int, v: int, channels: int = 3) -> np.ndarray:
    return np.full(w * h * channels, v, np.uint8)


def photo(w: int, h: int, seed: int = 42) -> np.ndarray:
    """A photograph-like synthetic (round 5): smooth structure at several scales + a little sensor noise — integer arithmetic
    only (box filters by cumulative sums over the lcg noise), so every platform makes the same bytes.  At q = 80, 4:2:0 it
    codes to roughly 1-2 bits per pixel, between the pure gradient (0.15) and noise (5.3) that bracket it in bench.py."""
    def box(a, k):  # k x k box sum of an int64 image, edge replicated, integer mean
        p = k // 2
        a = np.pad(a, ((p, p), (p, p), (0, 0)), mode="edge")
        c = np.cumsum(a, axis=0, dtype=np.int64)
        a = c[k - 1:] - np.concatenate([np.zeros((1,) + c.shape[1:], np.int64), c[:-k]], axis=0)
        c = np.cumsum(a, axis=1, dtype=np.int64)
        a = c[:, k - 1:] - np.concatenate([np.zeros((c.shape[0], 1, c.shape[2]), np.int64), c[:, :-k]], axis=1)
        return a // (k * k)
    n = lcg_bytes(w * h * 3, seed).reshape(h, w, 3).astype(np.int64)
    coarse = box(box(n, 31), 31)          # large structures
    mid = box(box(n[::-1, ::-1], 9), 9)   # medium detail
    fine = box(n[:, ::-1], 3)             # texture
    v = 128 + (coarse - 128) * 14 + (mid - 128) * 3 + (fine - 128) // 3 + (n - 128) // 40
    lum = v.sum(axis=2, keepdims=True) // 3   # correlated colour: mostly luminance, a little chroma
    v = lum + (v - lum) // 3
    return np.clip(v, 0, 255).astype(np.uint8).reshape(-1)
