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


def scene(w: int, h: int, seed: int = 42) -> np.ndarray:
    """A second photograph-like synthetic (round 6), built differently from `photo` (sums of blurred noise): what
    a camera sees of a man-made scene — FLAT regions with HARD edges (a posterised coarse field picks each pixel's region and the
    region its colour from a palette that includes saturated primaries), oriented fine TEXTURE inside the regions (hatching whose
    direction, period and contrast change per region), a smooth illumination gradient across the frame, small SATURATED chroma
    details (spots of pure red / blue / yellow / green where a mid-scale field peaks) and a little sensor noise.  Integer
    arithmetic only (box filters by cumulative sums over the lcg bytes): every platform makes the same bytes."""
    def box(a, k):  # k x k box mean of an int64 plane, edge replicated
        p = k // 2
        a = np.pad(a, ((p, p), (p, p)), mode="edge")
        c = np.cumsum(a, axis=0, dtype=np.int64)
        a = c[k - 1:] - np.concatenate([np.zeros((1, c.shape[1]), np.int64), c[:-k]], axis=0)
        c = np.cumsum(a, axis=1, dtype=np.int64)
        a = c[:, k - 1:] - np.concatenate([np.zeros((c.shape[0], 1), np.int64), c[:, :-k]], axis=1)
        return a // (k * k)
    n = lcg_bytes(w * h * 3, seed).reshape(h, w, 3).astype(np.int64)
    yy, xx = np.arange(h, dtype=np.int64)[:, None], np.arange(w, dtype=np.int64)[None, :]
    kc = 41 if min(w, h) >= 128 else 9
    coarse = box(box(n[:, :, 0], kc), kc)                 # ~N(128, small): stretch, then posterise into regions
    lo, hi = int(coarse.min()), int(coarse.max())
    region = ((coarse - lo) * 11 // max(hi - lo, 1)).clip(0, 10)   # 11 bands -> contiguous regions with hard borders
    palette = np.array([[32, 36, 44], [200, 196, 180], [224, 32, 24], [40, 96, 200], [236, 212, 40], [60, 150, 70], [120, 120, 124],
                        [250, 250, 250], [90, 50, 30], [180, 60, 160], [16, 16, 16]], np.int64)
    v = palette[region]                                    # (h, w, 3)
    # oriented texture: per region a direction (a, b), a period and a contrast
    a = np.array([1, 3, 0, 2, 5, 1, 7, 2, 1, 4, 3], np.int64)[region]
    b = np.array([0, 1, 1, 5, 2, 6, 1, 3, 1, 1, 4], np.int64)[region]
    per = np.array([4, 6, 3, 8, 5, 7, 3, 12, 4, 6, 5], np.int64)[region]
    amp = np.array([10, 18, 6, 24, 12, 30, 4, 8, 20, 14, 3], np.int64)[region]
    phase = (xx * a + yy * b) % (2 * per)
    tri = np.where(phase < per, phase, 2 * per - phase)    # triangle wave 0..per
    v = v + ((tri * 2 - per) * amp // per)[:, :, None]
    # illumination: darker towards one corner
    v = v * (160 + (xx * 60) // max(w - 1, 1) + (yy * 36) // max(h - 1, 1))[:, :, None] // 256 + 20
    # saturated details where a mid-scale field peaks
    km = 7 if min(w, h) >= 32 else 3
    mid = box(n[::-1, :, 1], km)
    top = int(np.sort(mid.reshape(-1))[max(0, mid.size - 1 - mid.size // 40)])   # the highest 2.5 %
    spots = mid >= top
    prim = np.array([[255, 0, 0], [0, 0, 255], [255, 255, 0], [0, 255, 0]], np.int64)[(n[:, :, 2] >> 2) & 3]
    blk = ((xx // 8) + (yy // 8)) % 4                       # (the colour is constant over 8 x 8 cells: spots, not confetti)
    prim = np.array([[255, 0, 0], [0, 0, 255], [255, 255, 0], [0, 255, 0]], np.int64)[blk + 0 * prim[:, :, 0]]
    v = np.where(spots[:, :, None], prim, v)
    v = v + (n - 128) // 24                                # sensor noise
    return np.clip(v, 0, 255).astype(np.uint8).reshape(-1)
