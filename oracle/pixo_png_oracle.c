/*
 * pixo_png_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's (leerob/pixo v0.4.1) PNG row-filter stage and of its
 * Adler-32: what `apply_filters` hands to DEFLATE (SURVEY §8 "PNG (config 5) path items", §8f-3).
 * Used only as the checker in tests/ and as bench.py's cpu_baseline for the c5 workload.
 *
 * Parity status: PINNED against the reference's own WebAssembly build: tests/golden/make_golden_png.py
 * runs `encodePng` (src/wasm.rs:79) under node, inflates the IDAT stream and records filter bytes,
 * filtered bytes (sha256; small cases verbatim) and the zlib trailer (= Adler-32 of the filtered
 * stream); tests/test_png_oracle_golden.py requires this file to reproduce all of them, including
 * the 4096x4096 RGBA case of SURVEY §8c.  The wasm build has no `parallel` feature: its
 * AdaptiveFast is the sequential, stateful variant (filter.rs:147-167); the stateless per-row
 * variant that the rayon path (and the GPU) runs shares every function below with it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "pixo_png_oracle.h"

/* simd/fallback.rs:8-25 */
uint32_t po_adler32(const uint8_t *data, size_t n)
{
    const uint32_t MOD = 65521u;
    const size_t NMAX = 5552;
    uint32_t s1 = 1, s2 = 0;
    for (size_t i = 0; i < n;) {
        size_t end = i + NMAX < n ? i + NMAX : n;
        for (; i < end; i++) { s1 += data[i]; s2 += s1; }
        s1 %= MOD; s2 %= MOD;
    }
    return (s2 << 16) | s1;
}

/* simd/fallback.rs:93 score_filter: sum of |b as i8| */
static uint64_t score(const uint8_t *f, size_t n)
{
    uint64_t s = 0;
    for (size_t i = 0; i < n; i++) { int v = (int8_t)f[i]; s += (uint64_t)(v < 0 ? -v : v); }
    return s;
}

/* simd/fallback.rs:102-159 */
static void f_sub(const uint8_t *row, size_t n, size_t bpp, uint8_t *out)
{
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(row[i] - (i >= bpp ? row[i - bpp] : 0));
}
static void f_up(const uint8_t *row, const uint8_t *prev, size_t n, uint8_t *out)
{
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(row[i] - prev[i]);
}
static void f_avg(const uint8_t *row, const uint8_t *prev, size_t n, size_t bpp, uint8_t *out)
{
    for (size_t i = 0; i < n; i++) {
        unsigned left = i >= bpp ? row[i - bpp] : 0, above = prev[i];
        out[i] = (uint8_t)(row[i] - (uint8_t)((left + above) / 2));
    }
}
static uint8_t paeth(uint8_t a8, uint8_t b8, uint8_t c8)
{
    int a = a8, b = b8, c = c8, p = a + b - c;
    int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a8;
    if (pb <= pc) return b8;
    return c8;
}
static void f_paeth(const uint8_t *row, const uint8_t *prev, size_t n, size_t bpp, uint8_t *out)
{
    for (size_t i = 0; i < n; i++) {
        uint8_t left = i >= bpp ? row[i - bpp] : 0, ul = i >= bpp ? prev[i - bpp] : 0;
        out[i] = (uint8_t)(row[i] - paeth(left, prev[i], ul));
    }
}

/* filter.rs:302-393 adaptive_filter (MinSum is an alias, :395-404).  out[0] = filter byte. */
static void adaptive(const uint8_t *row, const uint8_t *prev, size_t n, size_t bpp, uint8_t *out, uint8_t *tmp)
{
    const uint64_t early = (uint64_t)n / 4 + 1;
    uint64_t best = UINT64_MAX, s;
    /* None */
    s = score(row, n);
    if (s < best) {
        best = s;
        out[0] = PO_F_NONE; memcpy(out + 1, row, n);
        if (best <= early) return;
    }
    if (best == 0) return;
    f_sub(row, n, bpp, tmp);
    s = score(tmp, n);
    if (s < best) { best = s; out[0] = PO_F_SUB; memcpy(out + 1, tmp, n); if (best == 0 || best <= early) return; }
    f_up(row, prev, n, tmp);
    s = score(tmp, n);
    if (s < best) { best = s; out[0] = PO_F_UP; memcpy(out + 1, tmp, n); if (best == 0 || best <= early) return; }
    f_avg(row, prev, n, bpp, tmp);
    s = score(tmp, n);
    if (s < best) { best = s; out[0] = PO_F_AVG; memcpy(out + 1, tmp, n); if (best == 0 || best <= early) return; }
    f_paeth(row, prev, n, bpp, tmp);
    s = score(tmp, n);
    if (s < best) { out[0] = PO_F_PAETH; memcpy(out + 1, tmp, n); }
}

/* filter.rs:474-527 adaptive_filter_fast */
static void adaptive_fast(const uint8_t *row, const uint8_t *prev, size_t n, size_t bpp, uint8_t *out, uint8_t *tmp)
{
    const uint64_t early = (uint64_t)n / 8 + 1;
    f_sub(row, n, bpp, out + 1);
    out[0] = PO_F_SUB;
    uint64_t best = score(out + 1, n);
    if (best <= early) return;
    f_up(row, prev, n, tmp);
    uint64_t s = score(tmp, n);
    if (s < best) { best = s; out[0] = PO_F_UP; memcpy(out + 1, tmp, n); }
    if (best <= early) return;
    f_paeth(row, prev, n, bpp, tmp);
    s = score(tmp, n);
    if (s < best) { out[0] = PO_F_PAETH; memcpy(out + 1, tmp, n); }
}

/* filter.rs:635-649 score_bigrams: number of distinct adjacent byte pairs of the filtered row */
static uint64_t score_bigrams(const uint8_t *f, size_t n)
{
    static _Thread_local uint8_t seen[65536];
    memset(seen, 0, sizeof seen);
    uint64_t count = 0;
    for (size_t i = 0; i + 1 < n; i++) {
        const unsigned key = ((unsigned)f[i] << 8) | f[i + 1];
        if (!seen[key]) { seen[key] = 1; count++; }
    }
    return count;
}

/* filter.rs:406-472 bigrams_filter: None, Sub, Up, Average, Paeth in this order, a later filter wins
 * only with a strictly smaller count, no early exit */
static void bigrams(const uint8_t *row, const uint8_t *prev, size_t n, size_t bpp, uint8_t *out, uint8_t *tmp)
{
    uint64_t best = score_bigrams(row, n), s; /* (< usize::MAX always) */
    out[0] = PO_F_NONE; memcpy(out + 1, row, n);
    f_sub(row, n, bpp, tmp);
    s = score_bigrams(tmp, n);
    if (s < best) { best = s; out[0] = PO_F_SUB; memcpy(out + 1, tmp, n); }
    f_up(row, prev, n, tmp);
    s = score_bigrams(tmp, n);
    if (s < best) { best = s; out[0] = PO_F_UP; memcpy(out + 1, tmp, n); }
    f_avg(row, prev, n, bpp, tmp);
    s = score_bigrams(tmp, n);
    if (s < best) { best = s; out[0] = PO_F_AVG; memcpy(out + 1, tmp, n); }
    f_paeth(row, prev, n, bpp, tmp);
    s = score_bigrams(tmp, n);
    if (s < best) { out[0] = PO_F_PAETH; memcpy(out + 1, tmp, n); }
}

/* filter.rs:529-574 filter_row */
static int filter_row(const uint8_t *row, const uint8_t *prev, size_t n, size_t bpp, int strategy, uint8_t *out, uint8_t *tmp)
{
    switch (strategy) {
    case PO_S_NONE: out[0] = PO_F_NONE; memcpy(out + 1, row, n); return 0;
    case PO_S_SUB: out[0] = PO_F_SUB; f_sub(row, n, bpp, out + 1); return 0;
    case PO_S_UP: out[0] = PO_F_UP; f_up(row, prev, n, out + 1); return 0;
    case PO_S_AVERAGE: out[0] = PO_F_AVG; f_avg(row, prev, n, bpp, out + 1); return 0;
    case PO_S_PAETH: out[0] = PO_F_PAETH; f_paeth(row, prev, n, bpp, out + 1); return 0;
    case PO_S_MINSUM:
    case PO_S_ADAPTIVE: adaptive(row, prev, n, bpp, out, tmp); return 0;
    case PO_S_ADAPTIVE_FAST: adaptive_fast(row, prev, n, bpp, out, tmp); return 0;
    case PO_S_BIGRAMS: bigrams(row, prev, n, bpp, out, tmp); return 0;
    default: return -1;
    }
}

/* filter.rs:64-206 apply_filters_with_row_bytes.  stateful_fast != 0 selects the sequential
 * AdaptiveFast variant (:147-167, what a build without rayon runs); 0 the per-row stateless one
 * (apply_filters_parallel :576-611 semantics, rows independent). */
int po_png_filter(const uint8_t *data, uint32_t width, uint32_t height, uint32_t bpp, int strategy, int stateful_fast,
                  uint8_t *out, uint32_t *adler)
{
    const size_t n = (size_t)width * bpp;
    if (width == 0 || height == 0 || bpp == 0) return -1;
    const uint64_t area = (uint64_t)width * height;
    if (area <= 4096 && (strategy == PO_S_ADAPTIVE || strategy == PO_S_ADAPTIVE_FAST || strategy == PO_S_BIGRAMS))
        strategy = PO_S_SUB; /* :76-86 */
    uint8_t *zero = calloc(n ? n : 1, 1), *tmp = malloc(n ? n : 1);
    int last_fast = -1, rc = 0;
    for (uint32_t y = 0; y < height && rc == 0; y++) {
        const uint8_t *row = data + (size_t)y * n, *prev = y ? row - n : zero;
        uint8_t *o = out + (size_t)y * (n + 1);
        int s = strategy;
        if (strategy == PO_S_ADAPTIVE_FAST && stateful_fast) { /* bias toward the previous winner */
            if (last_fast == PO_F_SUB) s = PO_S_SUB;
            else if (last_fast == PO_F_UP) s = PO_S_UP;
            else if (last_fast == PO_F_PAETH) s = PO_S_PAETH;
        }
        rc = filter_row(row, prev, n, bpp, s, o, tmp);
        if (strategy == PO_S_ADAPTIVE_FAST) last_fast = o[0];
    }
    free(zero); free(tmp);
    if (rc == 0 && adler) *adler = po_adler32(out, (size_t)height * (n + 1)); /* deflate.rs:1044: once over the whole stream */
    return rc;
}
