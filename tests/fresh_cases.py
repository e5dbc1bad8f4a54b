"""Randomised JPEG cases for the differential campaign (tools/oracle_vs_wasm.py) and its committed record
(tests/golden/jpeg_fresh_cases.json): a case is ONE integer; content, size and options follow from it by integer arithmetic on
the raw 64-bit output of numpy's PCG64 (`random_raw`, whose stream numpy specifies), so that the build container (which runs the
reference's wasm), the CPU tests and the GPU box regenerate exactly the same bytes.  Content kinds aim at what the lcg noise /
gradient / flat-block generators of synth.py do not reach: amplitudes of a few grey levels around 128 (coefficients next to the
quantiser's rounding points), box-blurred noise ("photographic"), hard edges at random positions, sparse spikes on a flat ground,
random flat patches, channel-correlated colour."""
import numpy as np

KINDS = ("noise", "lowamp", "smooth", "patches", "sparse", "edges", "blur", "corr")


class _Raw:
    def __init__(self, seed):
        self.g = np.random.PCG64(seed)

    def u64(self, n):
        return self.g.random_raw(n).astype(np.uint64)

    def bytes(self, n):
        return self.u64((n + 7) // 8).view(np.uint8)[:n].copy()

    def below(self, k):
        return int(self.g.random_raw(1)[0] % np.uint64(k))


def _box_blur(a, r):
    """Integer box blur with edge replication, (2r+1)^2 window, floor division."""
    h, w = a.shape
    p = np.pad(a.astype(np.int64), r, mode="edge")
    c = np.zeros((h + 2 * r + 1, w + 2 * r + 1), np.int64)
    c[1:, 1:] = p.cumsum(0).cumsum(1)
    k = 2 * r + 1
    s = c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]
    return (s // (k * k)).astype(np.int64)


def content(kind, w, h, ch, raw):
    n = w * h
    if kind == "noise":
        return raw.bytes(n * ch)
    if kind == "lowamp":
        amp = 1 + raw.below(4)
        base = 96 + raw.below(64)
        return (base + (raw.bytes(n * ch).astype(np.int64) % (2 * amp + 1)) - amp).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.int64)
    if kind == "smooth":
        out = np.empty((h, w, ch), np.int64)
        for c in range(ch):
            a, b, d = raw.below(512) - 256, raw.below(512) - 256, raw.below(256)
            out[:, :, c] = d + (xx * a) // max(w, 1) + (yy * b) // max(h, 1)
        out += raw.bytes(n * ch).reshape(h, w, ch).astype(np.int64) % (1 + raw.below(4))
        return np.clip(out, 0, 255).astype(np.uint8).reshape(-1)
    if kind == "patches":
        s = 3 + raw.below(30)
        gh, gw = h // s + 1, w // s + 1
        pal = raw.bytes(gh * gw * ch).reshape(gh, gw, ch)
        out = pal[yy // s, xx // s].astype(np.int64)
        spikes = (raw.u64(n).reshape(h, w) % np.uint64(97)) == 0
        out[spikes] = 255 - out[spikes]
        return out.astype(np.uint8).reshape(-1)
    if kind == "sparse":
        ground = raw.below(256)
        out = np.full((h, w, ch), ground, np.int64)
        hit = (raw.u64(n).reshape(h, w) % np.uint64(50 + raw.below(400))) == 0
        vals = raw.bytes(n * ch).reshape(h, w, ch)
        out[hit] = vals[hit]
        return out.astype(np.uint8).reshape(-1)
    if kind == "edges":
        period = 2 + raw.below(23)
        phase = raw.below(period)
        diag = raw.below(3)
        t = (xx + (yy * diag) + phase) // period
        lo, hi = raw.below(40), 215 + raw.below(41)
        out = np.where((t & 1)[:, :, None] == 0, lo, hi) + np.zeros((1, 1, ch), np.int64)
        if ch == 3:
            out[:, :, raw.below(3)] = 255 - out[:, :, raw.below(3)]
        return out.astype(np.uint8).reshape(-1)
    if kind == "blur":
        r = 1 + raw.below(4)
        out = np.empty((h, w, ch), np.int64)
        for c in range(ch):
            b = _box_blur(raw.bytes(n).reshape(h, w), r)
            out[:, :, c] = 128 + (b - 128) * (2 + raw.below(4))
        return np.clip(out, 0, 255).astype(np.uint8).reshape(-1)
    if kind == "corr":
        luma = _box_blur(raw.bytes(n).reshape(h, w), 1 + raw.below(2))
        out = np.empty((h, w, ch), np.int64)
        for c in range(ch):
            out[:, :, c] = luma + (raw.bytes(n).reshape(h, w).astype(np.int64) % 9) - 4 + raw.below(30) - 15
        return np.clip(out, 0, 255).astype(np.uint8).reshape(-1)
    raise ValueError(kind)


def case_of(case_id, max_side=320):
    """(dict of options, pixel bytes) of case `case_id`."""
    raw = _Raw(0x9E3779B97F4A7C15 ^ (case_id * 0x100000001B3))
    kind = KINDS[raw.below(len(KINDS))]
    shape = raw.below(10)
    if shape == 0:
        w, h = 1 + raw.below(16), 1 + raw.below(16)
    elif shape == 1:
        w, h = 1 + raw.below(max_side * 3), 1 + raw.below(24)
    elif shape == 2:
        w, h = 1 + raw.below(24), 1 + raw.below(max_side * 3)
    else:
        w, h = 1 + raw.below(max_side), 1 + raw.below(max_side)
    color_type = 0 if raw.below(4) == 0 else 2
    qsel = raw.below(6)
    quality = (1 + raw.below(100)) if qsel < 4 else ((90 + raw.below(11)) if qsel == 4 else (1 + raw.below(15)))
    preset = (0, 0, 1, 2)[raw.below(4)]
    s420 = raw.below(2) == 1
    ch = 1 if color_type == 0 else 3
    px = content(kind, w, h, ch, raw)
    assert px.size == w * h * ch and px.dtype == np.uint8
    return dict(id=case_id, kind=kind, w=w, h=h, color_type=color_type, quality=quality, preset=preset, s420=s420), px


PNG_BPP = {0: 1, 1: 2, 2: 3, 3: 4}


def png_case_of(case_id, max_side=200):
    """(dict, pixel bytes) of PNG row-filter case `case_id`: colour types 0..3 (gray, gray+alpha, RGB, RGBA), wasm presets 0..2
    (AdaptiveFast / Adaptive / Bigrams).  Alpha samples are made odd (never 0: the reference's optimize_alpha rewrites pixels under
    alpha 0 at presets 1 and 2).  Presets 1 and 2 also reduce colour type / palettise when the content allows: the campaign
    skips such cases (the PNG's IHDR says so), the recorded ones are those the reference left in their pixel format."""
    raw = _Raw(0xC2B2AE3D27D4EB4F ^ (case_id * 0x100000001B3))
    kind = KINDS[raw.below(len(KINDS))]
    ct = raw.below(4)
    preset = (0, 0, 1, 1, 2)[raw.below(5)]
    side = 48 if preset == 2 else max_side            # preset 2 runs the reference's optimal DEFLATE: small images only
    shape = raw.below(8)
    if shape == 0:
        w, h = 1 + raw.below(12), 1 + raw.below(12)
    elif shape == 1:
        w, h = 1 + raw.below(side * 4), 1 + raw.below(8)
    else:
        w, h = 1 + raw.below(side), 1 + raw.below(side)
    ch = PNG_BPP[ct]
    px = content(kind, w, h, ch, raw)
    if ct in (1, 3):
        px[ch - 1::ch] |= 1
    return dict(id=case_id, kind=kind, w=w, h=h, color_type=ct, preset=preset), px
