"""Stateless AdaptiveFast (the reference's rayon path, `adaptive_filter_fast`, src/png/filter.rs:474-527) derived a SECOND
time, by another route than oracle/pixo_png_oracle.c: the wasm build cannot produce this strategy's choices (it has no
rayon, so its AdaptiveFast is the stateful sequential variant), which left the filter CHOICES of the stateless variant
pinned only by the C restatement.  Here they are re-derived with whole-image numpy arithmetic — all three
candidate planes at once, scores as matrix row sums, the decision as boolean algebra over the score vectors instead of the
reference's early-return control flow — and the C oracle must pick the same filter for every row and emit the same bytes.
The Paeth plane is computed from the predictor's distance definition, (src/simd/fallback.rs:142-159), with int64 numpy
arrays rather than the oracle's per-byte branches."""
import numpy as np
import pytest

import oracle_lib as O


def planes(img, bpp):
    """(sub, up, paeth) filtered planes of an h x row_bytes uint8 image, int64 arithmetic reduced mod 256."""
    x = img.astype(np.int64)
    h, n = x.shape
    b = np.vstack([np.zeros((1, n), np.int64), x[:-1]])                        # above (zero row for y = 0)
    a = np.hstack([np.zeros((h, bpp), np.int64), x[:, :-bpp]]) if n > bpp else np.zeros_like(x)   # left
    c = np.hstack([np.zeros((h, bpp), np.int64), b[:, :-bpp]]) if n > bpp else np.zeros_like(x)    # upper left
    p = a + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
    return [((x - a) % 256).astype(np.uint8), ((x - b) % 256).astype(np.uint8), ((x - pred) % 256).astype(np.uint8)]


def score(plane):
    """sum over a row of |byte as i8| (fallback.rs:93): 0..127 as they are, 128..255 count 256 - v"""
    v = plane.astype(np.int64)
    return np.where(v < 128, v, 256 - v).sum(axis=1)


def adaptive_fast_stateless(img, bpp):
    sub, up, paeth = planes(img, bpp)
    s_sub, s_up, s_pae = score(sub), score(up), score(paeth)
    early = img.shape[1] // 8 + 1
    # the reference's sequence as algebra: Sub stays unless ... (strict '<' keeps the earlier filter)
    stop_after_sub = s_sub <= early
    best_after_up = np.where(s_up < s_sub, s_up, s_sub)
    up_wins = (s_up < s_sub) & ~stop_after_sub
    stop_after_up = ~stop_after_sub & (best_after_up <= early)
    paeth_wins = ~stop_after_sub & ~stop_after_up & (s_pae < best_after_up)
    choice = np.where(paeth_wins, 4, np.where(up_wins, 2, 1))
    out = np.empty((img.shape[0], img.shape[1] + 1), np.uint8)
    out[:, 0] = choice
    for f, pl in ((1, sub), (2, up), (4, paeth)):
        rows = choice == f
        out[rows, 1:] = pl[rows]
    return out


def images():
    rng = np.random.RandomState(7)
    for (w, h, bpp) in [(120, 40, 3), (80, 64, 4), (200, 33, 1), (150, 70, 2), (60, 90, 6), (50, 120, 8), (512, 48, 4)]:
        n = w * bpp
        yield "noise", w, h, bpp, rng.randint(0, 256, (h, n)).astype(np.uint8)
        yy, xx = np.mgrid[0:h, 0:n]
        yield "gradient", w, h, bpp, ((xx // bpp * 3 + yy * 5) % 256).astype(np.uint8)                  # Sub / Up rows
        yield "diagonal", w, h, bpp, (((xx // bpp + yy) * 9 + (xx % bpp) * 40) % 256).astype(np.uint8)    # Paeth rows
        flat = np.full((h, n), 77, np.uint8); flat[::7] = rng.randint(0, 256, (len(range(0, h, 7)), n)).astype(np.uint8)
        yield "flat+noise rows", w, h, bpp, flat                                                           # early stops
        small = rng.randint(0, 5, (h, n)).astype(np.int64); small[h // 2:] = (small[h // 2:] * 60) % 256; small = small.astype(np.uint8)
        yield "near the early-stop threshold", w, h, bpp, small


@pytest.mark.parametrize("case", list(images()), ids=lambda c: "%s_%dx%d_bpp%d" % (c[0], c[1], c[2], c[3]))
def test_stateless_adaptive_fast_choices_rederived(case):
    kind, w, h, bpp, img = case
    assert h > 32 and w * h > 4096  # (the reference takes its rayon path only above 32 rows; <= 4096 pixels are forced to Sub)
    want = adaptive_fast_stateless(img, bpp)
    got, adler = O.png_filter(img.reshape(-1), w, h, bpp, O.S_ADAPTIVE_FAST, False)
    got = got.reshape(h, w * bpp + 1)
    assert "".join(map(str, got[:, 0])) == "".join(map(str, want[:, 0])), kind
    assert np.array_equal(got, want)
    import zlib
    assert adler == zlib.adler32(want.tobytes())


def test_the_rederivation_exercises_every_branch():
    seen = set()
    for kind, w, h, bpp, img in images():
        seen |= set(adaptive_fast_stateless(img, bpp)[:, 0].tolist())
    assert seen == {1, 2, 4}
