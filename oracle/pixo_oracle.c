/*
 * pixo_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see pixo_oracle.h).
 *
 * CPU restatement of leerob/pixo v0.4.1's baseline JPEG encoder.  Each function
 * cites the reference file:line it follows.  Parity PINNED against the
 * reference's wasm build (tests/test_oracle_golden.py).
 *
 * Compile with -ffp-contract=off and without -ffast-math: the reference's
 * f32 arithmetic is one IEEE rounding per operation.
 */
#include "pixo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* colour: src/color.rs:60-77                                                 */
/* ------------------------------------------------------------------------- */
static inline int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

void po_rgb_to_ycbcr(uint8_t r8, uint8_t g8, uint8_t b8, uint8_t out[3])
{
    int r = r8, g = g8, b = b8;
    /* `>>` on i32 in Rust is an arithmetic shift; gcc's is too for signed int. */
    int y = (77 * r + 150 * g + 29 * b + 128) >> 8;
    int cb = ((-43 * r - 85 * g + 128 * b + 128) >> 8) + 128;
    int cr = ((128 * r - 107 * g - 21 * b + 128) >> 8) + 128;
    out[0] = (uint8_t)clamp255(y);
    out[1] = (uint8_t)clamp255(cb);
    out[2] = (uint8_t)clamp255(cr);
}

/* ------------------------------------------------------------------------- */
/* quantisation tables: src/jpeg/quantize.rs:4-89                             */
/* ------------------------------------------------------------------------- */
static const uint8_t STD_LUM[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t STD_CHR[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

/* quantize.rs:18-22 */
const uint8_t PO_ZIGZAG[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static inline uint32_t scaled_q(uint8_t base, uint32_t scale)
{
    uint32_t v = ((uint32_t)base * scale + 50u) / 100u;
    if (v < 1) v = 1;
    if (v > 255) v = 255;
    return v;
}

void po_quant_tables(uint8_t quality, uint8_t lum_zz[64], uint8_t chr_zz[64],
                     float lum_nat[64], float chr_nat[64])
{
    uint32_t q = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    uint32_t scale = q < 50 ? 5000u / q : 200u - 2u * q; /* quantize.rs:46-50 */
    for (int i = 0; i < 64; i++) {
        lum_zz[i] = (uint8_t)scaled_q(STD_LUM[PO_ZIGZAG[i]], scale);
        chr_zz[i] = (uint8_t)scaled_q(STD_CHR[PO_ZIGZAG[i]], scale);
        lum_nat[i] = (float)scaled_q(STD_LUM[i], scale);
        chr_nat[i] = (float)scaled_q(STD_CHR[i], scale);
    }
}

/* ------------------------------------------------------------------------- */
/* f32 AAN DCT: src/jpeg/dct.rs:591-700                                       */
/* ------------------------------------------------------------------------- */
#define PO_A1 0.70710678118654752440f /* FRAC_1_SQRT_2 */
#define PO_A2 0.5411961f
#define PO_A3 0.70710678118654752440f
#define PO_A4 1.3065629f
#define PO_A5 0.38268343f
static const float PO_S[8] = {0.3535534f, 0.2548978f, 0.2705981f, 0.3006724f,
                              0.3535534f, 0.4499881f, 0.6532815f, 1.2814578f};

/* dct.rs:651-700.  volatile-free: relies on -ffp-contract=off. */
static inline void aan_1d(float d[8])
{
    float t0 = d[0] + d[7], t7 = d[0] - d[7];
    float t1 = d[1] + d[6], t6 = d[1] - d[6];
    float t2 = d[2] + d[5], t5 = d[2] - d[5];
    float t3 = d[3] + d[4], t4 = d[3] - d[4];

    float e0 = t0 + t3, e3 = t0 - t3;
    float e1 = t1 + t2, e2 = t1 - t2;

    d[0] = e0 + e1;
    d[4] = e0 - e1;
    float z1 = (e2 + e3) * PO_A1;
    d[2] = e3 + z1;
    d[6] = e3 - z1;

    float o0 = t4 + t5, o1 = t5 + t6, o2 = t6 + t7;
    float z5 = (o0 - o2) * PO_A5;
    float z2 = o0 * PO_A2 + z5;
    float z4 = o2 * PO_A4 + z5;
    float z3 = o1 * PO_A3;
    float z11 = t7 + z3, z13 = t7 - z3;

    d[5] = z13 + z2;
    d[3] = z13 - z2;
    d[1] = z11 + z4;
    d[7] = z11 - z4;

    for (int i = 0; i < 8; i++) d[i] *= PO_S[i]; /* dct.rs:697-699 */
}

/* dct.rs:614-643: rows first, then columns. */
void po_dct_2d(const float in[64], float out[64])
{
    float tmp[64], v[8];
    for (int r = 0; r < 8; r++) {
        memcpy(v, in + r * 8, sizeof v);
        aan_1d(v);
        memcpy(tmp + r * 8, v, sizeof v);
    }
    for (int c = 0; c < 8; c++) {
        for (int r = 0; r < 8; r++) v[r] = tmp[r * 8 + c];
        aan_1d(v);
        for (int r = 0; r < 8; r++) out[r * 8 + c] = v[r];
    }
}

/* quantize.rs:99-105: IEEE divide, f32::round (half away from zero), `as i16` saturates. */
void po_quantize_block(const float dct[64], const float q[64], int16_t out[64])
{
    for (int i = 0; i < 64; i++) {
        float r = roundf(dct[i] / q[i]);
        if (r > 32767.0f) r = 32767.0f;
        if (r < -32768.0f) r = -32768.0f;
        out[i] = (int16_t)r;
    }
}

void po_zigzag(const int16_t in[64], int16_t out[64])
{
    for (int i = 0; i < 64; i++) out[i] = in[PO_ZIGZAG[i]];
}

/* ------------------------------------------------------------------------- */
/* block extraction: src/jpeg/mod.rs:1565-1656                                */
/* ------------------------------------------------------------------------- */
static inline size_t minsz(size_t a, size_t b) { return a < b ? a : b; }

/* jpeg/mod.rs:1565-1606 */
static void extract_block(const uint8_t *data, size_t w, size_t h, size_t bx, size_t by,
                          int color_type, float yb[64], float cbb[64], float crb[64])
{
    for (size_t dy = 0; dy < 8; dy++) {
        for (size_t dx = 0; dx < 8; dx++) {
            size_t x = minsz(bx + dx, w - 1), y = minsz(by + dy, h - 1);
            size_t i = dy * 8 + dx;
            if (color_type == PO_GRAY) {
                yb[i] = (float)data[y * w + x] - 128.0f;
                cbb[i] = 0.0f;
                crb[i] = 0.0f;
            } else {
                const uint8_t *p = data + (y * w + x) * 3;
                uint8_t c[3];
                po_rgb_to_ycbcr(p[0], p[1], p[2], c);
                yb[i] = (float)c[0] - 128.0f;
                cbb[i] = (float)c[1] - 128.0f;
                crb[i] = (float)c[2] - 128.0f;
            }
        }
    }
}

/* jpeg/mod.rs:1608-1656: f32 accumulation of the (already u8-rounded) chroma
 * over each 2x2 quad, then *0.25 - 128. */
static void extract_mcu_420(const uint8_t *data, size_t w, size_t h, size_t mx, size_t my,
                            float yb[4][64], float cbb[64], float crb[64])
{
    memset(cbb, 0, 64 * sizeof(float));
    memset(crb, 0, 64 * sizeof(float));
    for (size_t by = 0; by < 2; by++)
        for (size_t bx = 0; bx < 2; bx++)
            for (size_t dy = 0; dy < 8; dy++)
                for (size_t dx = 0; dx < 8; dx++) {
                    size_t gx = bx * 8 + dx, gy = by * 8 + dy;
                    size_t x = minsz(mx + gx, w - 1), y = minsz(my + gy, h - 1);
                    const uint8_t *p = data + (y * w + x) * 3;
                    uint8_t c[3];
                    po_rgb_to_ycbcr(p[0], p[1], p[2], c);
                    yb[by * 2 + bx][dy * 8 + dx] = (float)c[0] - 128.0f;
                    size_t ci = (gy / 2) * 8 + gx / 2;
                    cbb[ci] += (float)c[1];
                    crb[ci] += (float)c[2];
                }
    for (int i = 0; i < 64; i++) {
        cbb[i] = cbb[i] * 0.25f - 128.0f;
        crb[i] = crb[i] * 0.25f - 128.0f;
    }
}

void po_coeff_geometry(uint32_t w, uint32_t h, uint8_t color_type, uint8_t subsampling,
                       size_t *y_blocks, size_t *c_blocks)
{
    if (color_type == PO_GRAY) {
        *y_blocks = (size_t)((w + 7) / 8) * ((h + 7) / 8);
        *c_blocks = 0;
    } else if (subsampling == PO_S444) {
        *y_blocks = (size_t)((w + 7) / 8) * ((h + 7) / 8);
        *c_blocks = *y_blocks;
    } else {
        size_t m = (size_t)((w + 15) / 16) * ((h + 15) / 16);
        *y_blocks = 4 * m;
        *c_blocks = m;
    }
}

/* ------------------------------------------------------------------------- */
/* trellis quantisation: src/jpeg/trellis.rs:18-299 (lambda = DEFAULT_LAMBDA)    */
/* ------------------------------------------------------------------------- */
static const uint8_t ZZ_NAT[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                   12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                   58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
typedef struct { float cost; uint8_t zero_run; uint16_t parent; int16_t value; } tstate;

static int16_t sat_i16(float v)
{ /* Rust `as i16`: saturating, NaN -> 0 */
    if (!(v == v)) return 0;
    if (v >= 32767.0f) return 32767;
    if (v <= -32768.0f) return -32768;
    return (int16_t)v;
}
static int tcat(int16_t v)
{ /* trellis.rs:289-296 */
    unsigned a = (unsigned)(v < 0 ? -(int)v : (int)v) & 0xFFFFu;
    int c = 0;
    while (a) { c++; a >>= 1; }
    return c;
}
static int gen_candidates(float fq, int16_t c[5])
{ /* trellis.rs:210-244: 0, floor, round, ceil, one step further out when |fq| > 1.5; no duplicates */
    int16_t r = sat_i16(roundf(fq)), fl = sat_i16(floorf(fq)), ce = sat_i16(ceilf(fq));
    int n = 0;
    c[n++] = 0;
#define HAS(v) ({ int f_ = 0; for (int i_ = 0; i_ < n; i_++) if (c[i_] == (v)) f_ = 1; f_; })
    if (fl != 0 && !HAS(fl)) c[n++] = fl;
    if (r != 0 && !HAS(r)) c[n++] = r;
    if (ce != 0 && !HAS(ce)) c[n++] = ce;
    if (fabsf(fq) > 1.5f) {
        int16_t ext = (int16_t)(fq >= 0.0f ? ce + 1 : fl - 1);
        if (!HAS(ext)) c[n++] = ext;
    }
#undef HAS
    return n;
}
static float ac_huff_len(unsigned rs)
{ /* trellis.rs:260-279 */
    switch (rs) {
    case 0x00: return 4.0f; case 0x01: return 2.0f; case 0x02: return 2.5f; case 0x03: return 3.0f;
    case 0x04: return 4.0f; case 0x11: return 3.0f; case 0x12: return 4.0f; case 0x21: return 4.0f;
    case 0xF0: return 10.0f;
    default: { float run = (float)(rs >> 4), size = (float)(rs & 0x0F); return 3.0f + run * 0.5f + size * 0.3f; }
    }
}
static void trellis_quantize(const float dct[64], const float q[64], int16_t out[64])
{ /* trellis.rs:67-208 */
    const float lambda = 1.0f;
    memset(out, 0, 64 * sizeof(int16_t));
    out[0] = sat_i16(roundf(dct[0] / q[0]));
    static _Thread_local tstate all[64][8];
    static _Thread_local int alln[64];
    tstate cur[8], nxt[40];
    int ncur = 1;
    cur[0].cost = 0.0f; cur[0].zero_run = 0; cur[0].parent = 0; cur[0].value = 0;
    all[0][0] = cur[0]; alln[0] = 1;
    for (int zz = 1; zz < 64; zz++) {
        const float coef = dct[ZZ_NAT[zz]], qq = q[ZZ_NAT[zz]];
        int16_t cand[5];
        const int nc = gen_candidates(coef / qq, cand);
        int nn = 0;
        for (int pi = 0; pi < ncur; pi++) {
            for (int ci = 0; ci < nc; ci++) {
                const int16_t c = cand[ci];
                const float rec = (float)c * qq, dd = coef - rec, dist = dd * dd;
                float rate; uint8_t nrun;
                if (c == 0) {
                    unsigned r = (unsigned)cur[pi].zero_run + 1; if (r > 255) r = 255;
                    if (r >= 16) { rate = 10.0f; nrun = 0; } else { rate = 0.0f; nrun = (uint8_t)r; }
                } else {
                    const int cat = tcat(c);
                    rate = ac_huff_len(((unsigned)cur[pi].zero_run << 4) | (unsigned)cat) + (float)cat;
                    nrun = 0;
                }
                const float cost = cur[pi].cost + rate + lambda * dist;
                int found = -1;
                for (int k = 0; k < nn; k++) if (nxt[k].value == c && nxt[k].zero_run == nrun) { found = k; break; }
                tstate st = {cost, nrun, (uint16_t)pi, c};
                if (found < 0) nxt[nn++] = st;
                else if (cost < nxt[found].cost) nxt[found] = st;
            }
        }
        /* stable sort by cost (sort_by + partial_cmp), keep the best MAX_STATES = 8 */
        for (int i = 1; i < nn; i++) {
            tstate t = nxt[i]; int j = i - 1;
            while (j >= 0 && nxt[j].cost > t.cost) { nxt[j + 1] = nxt[j]; j--; }
            nxt[j + 1] = t;
        }
        if (nn > 8) nn = 8;
        for (int i = 0; i < nn; i++) { cur[i] = nxt[i]; all[zz][i] = nxt[i]; }
        ncur = nn; alln[zz] = nn;
    }
    for (int i = 0; i < ncur; i++) if (cur[i].zero_run > 0) cur[i].cost += 4.0f;
    int best = 0;
    for (int i = 1; i < ncur; i++) if (cur[i].cost < cur[best].cost) best = i; /* min_by: the first minimum */
    int idx = best;
    for (int zz = 63; zz >= 1; zz--) {
        if (idx < alln[zz]) { out[ZZ_NAT[zz]] = all[zz][idx].value; idx = all[zz][idx].parent; }
    }
}

void po_trellis_quantize(const float dct[64], const float q[64], int16_t out[64]) { trellis_quantize(dct, q, out); }

static void dct_quant_ex(const float blk[64], const float q[64], int16_t out[64], int use_trellis)
{ /* quantize_dct, jpeg/mod.rs:968-976 */
    float f[64];
    po_dct_2d(blk, f);
    if (use_trellis) trellis_quantize(f, q, out);
    else po_quantize_block(f, q, out);
}
#define dct_quant(blk, q, out) dct_quant_ex(blk, q, out, use_trellis)

int po_jpeg_coeffs_ex(const uint8_t *pixels, uint32_t w32, uint32_t h32, uint8_t color_type,
                      uint8_t subsampling, uint8_t quality, int16_t *y, int16_t *cb,
                      int16_t *cr, int threads, int use_trellis);
int po_jpeg_coeffs(const uint8_t *pixels, uint32_t w32, uint32_t h32, uint8_t color_type,
                   uint8_t subsampling, uint8_t quality, int16_t *y, int16_t *cb,
                   int16_t *cr, int threads)
{
    return po_jpeg_coeffs_ex(pixels, w32, h32, color_type, subsampling, quality, y, cb, cr, threads, 0);
}

int po_jpeg_coeffs_ex(const uint8_t *pixels, uint32_t w32, uint32_t h32, uint8_t color_type,
                      uint8_t subsampling, uint8_t quality, int16_t *y, int16_t *cb,
                      int16_t *cr, int threads, int use_trellis)
{
    if (w32 == 0 || h32 == 0) return PO_ERR_INVALID_DIMENSIONS;
    if (color_type != PO_GRAY && color_type != PO_RGB) return PO_ERR_UNSUPPORTED_COLOR;
    size_t w = w32, h = h32;
    uint8_t lzz[64], czz[64];
    float ql[64], qc[64];
    po_quant_tables(quality, lzz, czz, ql, qc);
    (void)threads;

    if (color_type == PO_GRAY || subsampling == PO_S444) {
        long bw = (long)((w + 7) / 8), bh = (long)((h + 7) / 8);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 1 ? threads : 1)
#endif
        for (long brow = 0; brow < bh; brow++) {
            float yb[64], cbb[64], crb[64];
            for (long bcol = 0; bcol < bw; bcol++) {
                size_t bi = (size_t)brow * bw + bcol;
                extract_block(pixels, w, h, (size_t)bcol * 8, (size_t)brow * 8, color_type, yb,
                              cbb, crb);
                dct_quant(yb, ql, y + bi * 64);
                if (color_type != PO_GRAY) {
                    dct_quant(cbb, qc, cb + bi * 64);
                    dct_quant(crb, qc, cr + bi * 64);
                }
            }
        }
    } else {
        long mw = (long)((w + 15) / 16), mh = (long)((h + 15) / 16);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 1 ? threads : 1)
#endif
        for (long mrow = 0; mrow < mh; mrow++) {
            float yb[4][64], cbb[64], crb[64];
            for (long mcol = 0; mcol < mw; mcol++) {
                size_t mi = (size_t)mrow * mw + mcol;
                extract_mcu_420(pixels, w, h, (size_t)mcol * 16, (size_t)mrow * 16, yb, cbb, crb);
                for (int k = 0; k < 4; k++) dct_quant(yb[k], ql, y + (mi * 4 + k) * 64);
                dct_quant(cbb, qc, cb + mi * 64);
                dct_quant(crb, qc, cr + mi * 64);
            }
        }
    }
    return PO_OK;
}

/* ------------------------------------------------------------------------- */
/* Huffman tables: src/jpeg/huffman.rs:17-62 (JPEG Annex K), :215-291         */
/* ------------------------------------------------------------------------- */
static const uint8_t DC_LUM_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t DC_CHR_BITS[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t AC_LUM_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125};
static const uint8_t AC_CHR_BITS[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119};
static const uint8_t AC_LUM_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61,
    0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52,
    0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25,
    0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45,
    0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64,
    0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
    0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8,
    0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t AC_CHR_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61,
    0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33,
    0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18,
    0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44,
    0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63,
    0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
    0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7,
    0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

typedef struct {
    uint8_t bits[16];
    uint8_t vals[256];
    int nvals;
    uint16_t code[256];
    uint8_t len[256];
} htable;

/* huffman.rs:264-291 (and :215-261 for the standard tables): canonical codes. */
static int build_codes(htable *t, int table_len)
{
    memset(t->code, 0, sizeof t->code);
    memset(t->len, 0, sizeof t->len);
    uint16_t code = 0;
    int vi = 0;
    for (int l = 0; l < 16; l++) {
        for (int k = 0; k < t->bits[l]; k++) {
            if (vi >= t->nvals) return 0;
            int sym = t->vals[vi];
            if (sym >= table_len) return 0;
            t->code[sym] = code;
            t->len[sym] = (uint8_t)(l + 1);
            vi++;
            code++;
        }
        code = (uint16_t)(code << 1);
    }
    return 1;
}

static void set_table(htable *t, const uint8_t bits[16], const uint8_t *vals, int n)
{
    memcpy(t->bits, bits, 16);
    memcpy(t->vals, vals, (size_t)n);
    t->nvals = n;
}

typedef struct {
    htable dc[2], ac[2]; /* [0] luminance, [1] chrominance */
} hufftables;

static void std_tables(hufftables *ht)
{
    set_table(&ht->dc[0], DC_LUM_BITS, DC_VALS, 12);
    set_table(&ht->dc[1], DC_CHR_BITS, DC_VALS, 12);
    set_table(&ht->ac[0], AC_LUM_BITS, AC_LUM_VALS, 162);
    set_table(&ht->ac[1], AC_CHR_BITS, AC_CHR_VALS, 162);
    build_codes(&ht->dc[0], 12);
    build_codes(&ht->dc[1], 12);
    build_codes(&ht->ac[0], 256);
    build_codes(&ht->ac[1], 256);
}

/* huffman.rs:317-391 build_code_lengths: Huffman tree with a min-heap keyed on
 * (freq, node_index); node indices are unique so pop order is fully determined and
 * a linear scan for the two smallest keys reproduces it.  Code length = depth+1. */
static int build_code_lengths(const uint64_t *counts, int n, uint8_t *lengths)
{
    uint64_t freq[512];
    int left[512], right[512], sym[512], alive[512];
    int nn = 0;
    memset(lengths, 0, (size_t)n);
    for (int s = 0; s < n; s++)
        if (counts[s]) {
            freq[nn] = counts[s];
            left[nn] = right[nn] = -1;
            sym[nn] = s;
            alive[nn] = 1;
            nn++;
        }
    if (nn == 0) return 0;
    if (nn == 1) { lengths[sym[0]] = 1; return 1; }
    int live = nn;
    while (live > 1) {
        int a = -1, b = -1;
        for (int i = 0; i < nn; i++) {
            if (!alive[i]) continue;
            if (a < 0 || freq[i] < freq[a]) { b = a; a = i; }
            else if (b < 0 || freq[i] < freq[b]) b = i;
        }
        /* ties: lower index first — the scan visits ascending indices and uses strict < */
        alive[a] = alive[b] = 0;
        freq[nn] = freq[a] + freq[b];
        left[nn] = a;
        right[nn] = b;
        sym[nn] = -1;
        alive[nn] = 1;
        nn++;
        live--;
    }
    int root = nn - 1;
    int stack_n[512], stack_d[512], sp = 0;
    stack_n[sp] = root; stack_d[sp] = 0; sp++;
    while (sp) {
        sp--;
        int idx = stack_n[sp], d = stack_d[sp];
        if (sym[idx] >= 0) {
            int len = d + 1;
            if (len > 16) return 0;
            lengths[sym[idx]] = (uint8_t)len;
        } else {
            stack_n[sp] = left[idx]; stack_d[sp] = d + 1; sp++;
            stack_n[sp] = right[idx]; stack_d[sp] = d + 1; sp++;
        }
    }
    return 1;
}

/* huffman.rs:294-315 */
int po_build_bits_vals(const uint64_t *counts, int n, uint8_t bits[16], uint8_t *vals, int *nvals)
{
    uint8_t lengths[256];
    if (!build_code_lengths(counts, n, lengths)) return 0;
    memset(bits, 0, 16);
    for (int s = 0; s < n; s++)
        if (lengths[s]) {
            if (lengths[s] > 16) return 0;
            bits[lengths[s] - 1]++;
        }
    int k = 0;
    for (int l = 1; l <= 16; l++)
        for (int s = 0; s < n; s++)
            if (lengths[s] == l) vals[k++] = (uint8_t)s;
    *nvals = k;
    return 1;
}

/* huffman.rs:167-205 optimized_from_counts + jpeg/mod.rs:380-390 unwrap_or_default */
static void optimized_tables(hufftables *ht, uint64_t dc[2][12], uint64_t ac[2][256], int has_chroma)
{
    hufftables o;
    uint8_t bits[16], vals[256];
    int nv;
    std_tables(&o);
    if (!po_build_bits_vals(dc[0], 12, bits, vals, &nv)) { std_tables(ht); return; }
    set_table(&o.dc[0], bits, vals, nv);
    if (!po_build_bits_vals(ac[0], 256, bits, vals, &nv)) { std_tables(ht); return; }
    set_table(&o.ac[0], bits, vals, nv);
    if (has_chroma) {
        if (po_build_bits_vals(dc[1], 12, bits, vals, &nv)) set_table(&o.dc[1], bits, vals, nv);
        if (po_build_bits_vals(ac[1], 256, bits, vals, &nv)) set_table(&o.ac[1], bits, vals, nv);
    }
    if (!build_codes(&o.dc[0], 12) || !build_codes(&o.dc[1], 12) || !build_codes(&o.ac[0], 256) ||
        !build_codes(&o.ac[1], 256)) {
        std_tables(ht);
        return;
    }
    *ht = o;
}

/* ------------------------------------------------------------------------- */
/* growable byte vector + MSB-first bit writer: src/bits.rs:195-293           */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint8_t *p;
    size_t n, cap;
    int oom;
} bytevec;

static void bv_reserve(bytevec *v, size_t extra)
{
    if (v->n + extra <= v->cap) return;
    size_t nc = v->cap ? v->cap : 4096;
    while (nc < v->n + extra) nc *= 2;
    uint8_t *np = (uint8_t *)realloc(v->p, nc);
    if (!np) { v->oom = 1; return; }
    v->p = np;
    v->cap = nc;
}
static inline void bv_push(bytevec *v, uint8_t b)
{
    if (v->n == v->cap) { bv_reserve(v, 1); if (v->oom) return; }
    v->p[v->n++] = b;
}
static void bv_put(bytevec *v, const void *src, size_t n)
{
    bv_reserve(v, n);
    if (v->oom) return;
    memcpy(v->p + v->n, src, n);
    v->n += n;
}
static void bv_be16(bytevec *v, unsigned x) { bv_push(v, (uint8_t)(x >> 8)); bv_push(v, (uint8_t)x); }

typedef struct {
    bytevec *out;
    uint8_t cur;
    int pos; /* free bits in cur: 8 down to 0 (bits.rs:198) */
} bitw;

static inline void bw_emit(bitw *w)
{ /* bits.rs:245-253 */
    bv_push(w->out, w->cur);
    if (w->cur == 0xFF) bv_push(w->out, 0x00);
    w->cur = 0;
    w->pos = 8;
}
static inline void bw_bits(bitw *w, uint32_t val, int nbits)
{ /* bits.rs:216-242 */
    int rem = nbits;
    while (rem > 0) {
        int take = rem < w->pos ? rem : w->pos;
        int shift = rem - take;
        uint32_t chunk = (val >> shift) & ((1u << take) - 1u);
        w->pos -= take;
        w->cur |= (uint8_t)(chunk << w->pos);
        rem -= take;
        if (w->pos == 0) bw_emit(w);
    }
}
static void bw_flush(bitw *w)
{ /* bits.rs:261-272: pad with 1s */
    if (w->pos < 8) {
        w->cur |= (uint8_t)((1u << w->pos) - 1u);
        bw_emit(w);
    }
}

/* ------------------------------------------------------------------------- */
/* per-block symbolisation + emission: src/jpeg/huffman.rs:394-481            */
/* ------------------------------------------------------------------------- */
static inline int category(int v)
{ /* huffman.rs:394-401: 16 - leading_zeros(|v| as u16) */
    unsigned a = (unsigned)(v < 0 ? -v : v) & 0xFFFFu;
    int c = 0;
    while (a) { c++; a >>= 1; }
    return c;
}
static inline uint32_t value_bits(int v, int cat)
{ /* huffman.rs:404-418 */
    uint32_t b = (uint32_t)(uint16_t)(int16_t)(v < 0 ? v - 1 : v);
    return b & ((1u << cat) - 1u);
}

static int16_t encode_block(bitw *w, const int16_t blk[64], int16_t prev_dc, const htable *dct,
                            const htable *act)
{
    int16_t zz[64];
    po_zigzag(blk, zz);
    int16_t dc = zz[0];
    int16_t diff = (int16_t)(dc - prev_dc); /* i16 wrapping in release builds */
    int cat = category(diff);
    bw_bits(w, dct->code[cat], dct->len[cat]);
    if (cat > 0) bw_bits(w, value_bits(diff, cat), cat);
    int run = 0;
    for (int k = 1; k < 64; k++) {
        int ac = zz[k];
        if (ac == 0) { run++; continue; }
        while (run >= 16) { bw_bits(w, act->code[0xF0], act->len[0xF0]); run -= 16; }
        int c = category(ac);
        int rs = ((run << 4) | c) & 0xFF;
        bw_bits(w, act->code[rs], act->len[rs]);
        bw_bits(w, value_bits(ac, c), c);
        run = 0;
    }
    if (run > 0) bw_bits(w, act->code[0], act->len[0]);
    return dc;
}

/* jpeg/mod.rs:826-860 count_block */
static int16_t count_block(const int16_t blk[64], int16_t prev_dc, uint64_t *dcc, uint64_t *acc)
{
    int16_t zz[64];
    po_zigzag(blk, zz);
    int16_t dc = zz[0];
    dcc[category((int16_t)(dc - prev_dc))]++;
    int run = 0;
    for (int k = 1; k < 64; k++) {
        int ac = zz[k];
        if (ac == 0) { run++; continue; }
        while (run >= 16) { acc[0xF0]++; run -= 16; }
        acc[((run << 4) | category(ac)) & 0xFF]++;
        run = 0;
    }
    if (run > 0) acc[0]++;
    return dc;
}

/* Walks the coefficient tuple in scan order (jpeg/mod.rs:1448-1557), calling
 * either the counter or the emitter.  Restart logic: :1423-1445 / :706-727. */
typedef struct {
    int counting;
    bitw *w;
    const hufftables *ht;
    uint64_t (*dc)[12];
    uint64_t (*ac)[256];
} walk_ctx;

static inline int16_t do_block(walk_ctx *c, const int16_t *blk, int16_t prev, int cls)
{
    if (c->counting) return count_block(blk, prev, c->dc[cls], c->ac[cls]);
    return encode_block(c->w, blk, prev, &c->ht->dc[cls], &c->ht->ac[cls]);
}

static void walk_scan(walk_ctx *c, const int16_t *y, const int16_t *cb, const int16_t *cr,
                      const po_options *o)
{
    size_t yb, cbn;
    po_coeff_geometry(o->width, o->height, o->color_type, o->subsampling, &yb, &cbn);
    int is420 = (o->color_type != PO_GRAY && o->subsampling == PO_S420);
    size_t total = is420 ? cbn : yb;
    int16_t py = 0, pcb = 0, pcr = 0;
    uint8_t rst = 0;
    for (size_t m = 0; m < total; m++) {
        if (o->color_type == PO_GRAY) {
            py = do_block(c, y + m * 64, py, 0);
        } else if (!is420) {
            py = do_block(c, y + m * 64, py, 0);
            pcb = do_block(c, cb + m * 64, pcb, 1);
            pcr = do_block(c, cr + m * 64, pcr, 1);
        } else {
            for (int k = 0; k < 4; k++) py = do_block(c, y + (m * 4 + k) * 64, py, 0);
            pcb = do_block(c, cb + m * 64, pcb, 1);
            pcr = do_block(c, cr + m * 64, pcr, 1);
        }
        uint32_t cnt = (uint32_t)(m + 1);
        if (o->has_restart && o->restart_interval > 0 && cnt % o->restart_interval == 0 &&
            cnt < (uint32_t)total) {
            if (!c->counting) {
                bw_flush(c->w);
                bv_push(c->w->out, 0xFF);
                bv_push(c->w->out, (uint8_t)(0xD0 + (rst & 7)));
                rst = (uint8_t)((rst + 1) & 7);
            }
            py = pcb = pcr = 0;
        }
    }
}

int po_symbol_histograms(const int16_t *y, const int16_t *cb, const int16_t *cr,
                         const po_options *opt, uint64_t dc[2][12], uint64_t ac[2][256])
{
    memset(dc, 0, sizeof(uint64_t) * 24);
    memset(ac, 0, sizeof(uint64_t) * 512);
    walk_ctx c = {1, NULL, NULL, dc, ac};
    walk_scan(&c, y, cb, cr, opt);
    return PO_OK;
}

/* ------------------------------------------------------------------------- */
/* headers: src/jpeg/mod.rs:449-648                                           */
/* ------------------------------------------------------------------------- */
static void put_dht(bytevec *v, uint8_t id, const htable *t)
{ /* :597-612 */
    bv_be16(v, 0xFFC4);
    bv_be16(v, (unsigned)(2 + 1 + 16 + t->nvals));
    bv_push(v, id);
    bv_put(v, t->bits, 16);
    bv_put(v, t->vals, (size_t)t->nvals);
}

static void put_headers(bytevec *v, const po_options *o, const uint8_t lzz[64],
                        const uint8_t czz[64], const hufftables *ht)
{ /* progressive (:397-405): SOF2 instead of SOF0, and the SOS headers are written per scan */
    static const uint8_t app0[] = {0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    bv_be16(v, 0xFFD8);       /* SOI :449 */
    bv_put(v, app0, sizeof app0); /* APP0 :457-481 */
    bv_be16(v, 0xFFDB); bv_be16(v, 67); bv_push(v, 0); bv_put(v, lzz, 64); /* DQT :484-496 */
    bv_be16(v, 0xFFDB); bv_be16(v, 67); bv_push(v, 1); bv_put(v, czz, 64);
    int nc = o->color_type == PO_GRAY ? 1 : 3;
    bv_be16(v, o->progressive ? 0xFFC2 : 0xFFC0);       /* SOF0 / SOF2 :498-571 */
    bv_be16(v, (unsigned)(8 + 3 * nc));
    bv_push(v, 8);
    bv_be16(v, o->height & 0xFFFF);
    bv_be16(v, o->width & 0xFFFF);
    bv_push(v, (uint8_t)nc);
    if (nc == 1) {
        bv_push(v, 1); bv_push(v, 0x11); bv_push(v, 0);
    } else {
        bv_push(v, 1); bv_push(v, o->subsampling == PO_S420 ? 0x22 : 0x11); bv_push(v, 0);
        bv_push(v, 2); bv_push(v, 0x11); bv_push(v, 1);
        bv_push(v, 3); bv_push(v, 0x11); bv_push(v, 1);
    }
    put_dht(v, 0x00, &ht->dc[0]); /* :573-585 order */
    put_dht(v, 0x01, &ht->dc[1]);
    put_dht(v, 0x10, &ht->ac[0]);
    put_dht(v, 0x11, &ht->ac[1]);
    if (o->has_restart) { bv_be16(v, 0xFFDD); bv_be16(v, 4); bv_be16(v, o->restart_interval); } /* :587-591 */
    if (o->progressive) return;
    bv_be16(v, 0xFFDA);       /* SOS :614-648 */
    bv_be16(v, (unsigned)(6 + 2 * nc));
    bv_push(v, (uint8_t)nc);
    bv_push(v, 1); bv_push(v, 0x00);
    if (nc == 3) { bv_push(v, 2); bv_push(v, 0x11); bv_push(v, 3); bv_push(v, 0x11); }
    bv_push(v, 0); bv_push(v, 63); bv_push(v, 0);
}

/* ------------------------------------------------------------------------- */
/* progressive scans: src/jpeg/mod.rs:872-927, :1248-1365; src/jpeg/progressive.rs */
/* ------------------------------------------------------------------------- */
/* progressive.rs:363-381: walks BITS/VALS; a symbol that is not in the table gets (0, 4) */
static void code_from_table(const htable *t, uint8_t sym, unsigned *code, int *len)
{
    unsigned c = 0;
    int vi = 0;
    for (int l = 0; l < 16; l++) {
        for (int k = 0; k < t->bits[l]; k++) {
            if (vi < t->nvals && t->vals[vi] == sym) { *code = c; *len = l + 1; return; }
            vi++; c++;
        }
        c <<= 1;
    }
    *code = 0; *len = 4;
}
static void flush_eob_run(bitw *w, uint16_t *eob_run, const htable *ac)
{ /* progressive.rs:313-345 */
    if (*eob_run == 0) return;
    unsigned temp = *eob_run; int nbits = 0;
    while (temp > 0) { temp >>= 1; nbits++; }
    nbits = nbits > 0 ? nbits - 1 : 0;
    unsigned code; int len;
    code_from_table(ac, (uint8_t)(nbits << 4), &code, &len);
    bw_bits(w, code, len);
    if (nbits > 0) bw_bits(w, (uint32_t)(*eob_run - (1u << nbits)), nbits);
    *eob_run = 0;
}
static void encode_ac_first(bitw *w, const int16_t blk[64], int ss, int se, int al, uint16_t *eob_run, const htable *ac)
{ /* progressive.rs:141-210 */
    int16_t zz[64];
    po_zigzag(blk, zz);
    int k = se;
    while (k >= ss && (zz[k] >> al) == 0) { if (k == ss) break; k--; }
    const int last = k;
    if (last == ss && (zz[ss] >> al) == 0) {
        *eob_run += 1;
        if (*eob_run == 0x7FFF) flush_eob_run(w, eob_run, ac);
        return;
    }
    if (*eob_run > 0) flush_eob_run(w, eob_run, ac);
    int zero_run = 0;
    for (k = ss; k <= last; k++) {
        const int coef = zz[k] >> al;
        if (coef == 0) { zero_run++; continue; }
        unsigned code; int len;
        while (zero_run >= 16) { code_from_table(ac, 0xF0, &code, &len); bw_bits(w, code, len); zero_run -= 16; }
        const int cat = category(coef);
        code_from_table(ac, (uint8_t)((zero_run << 4) | cat), &code, &len);
        bw_bits(w, code, len);
        if (cat > 0) bw_bits(w, value_bits(coef, cat), cat);
        zero_run = 0;
    }
    if (last < se) *eob_run = 1;
}
static void progressive_scans(bytevec *v, const int16_t *y, size_t yb, const int16_t *cb, const int16_t *cr, size_t cbn,
                              const hufftables *ht)
{ /* simple_progressive_script (progressive.rs:98-110): DC of Y, Cb, Cr; Y AC 1-10, 11-63; Cb AC; Cr AC.
     Every scan walks the component's blocks in STORAGE order (mod.rs:1286, :1350) with its own bit writer. */
    static const struct { int comp, ss, se; } script[7] = {{0, 0, 0}, {1, 0, 0}, {2, 0, 0}, {0, 1, 10}, {0, 11, 63}, {1, 1, 63}, {2, 1, 63}};
    for (int s = 0; s < 7; s++) {
        const int comp = script[s].comp, ss = script[s].ss, se = script[s].se;
        bv_be16(v, 0xFFDA); bv_be16(v, 8); bv_push(v, 1); /* write_sos_progressive :650-682 */
        bv_push(v, (uint8_t)(comp + 1)); bv_push(v, comp == 0 ? 0x00 : 0x11);
        bv_push(v, (uint8_t)ss); bv_push(v, (uint8_t)se); bv_push(v, 0);
        const int16_t *coef = comp == 0 ? y : (comp == 1 ? cb : cr);
        const size_t n = comp == 0 ? yb : cbn;
        const int cls = comp == 0 ? 0 : 1;
        bitw w = {v, 0, 8};
        if (n == 0) continue; /* gray: the chroma scans are headers only */
        if (ss == 0 && se == 0) { /* encode_dc_scan :1248-1310 */
            int16_t prev = 0;
            for (size_t b = 0; b < n; b++) {
                const int16_t dc = coef[b * 64];
                const int16_t diff = (int16_t)(dc - prev);
                const int cat = category(diff);
                unsigned code; int len;
                code_from_table(&ht->dc[cls], (uint8_t)cat, &code, &len);
                bw_bits(&w, code, len);
                if (cat > 0) bw_bits(&w, value_bits(diff, cat), cat);
                prev = dc;
            }
        } else { /* encode_ac_first_scan :1326-1365 */
            uint16_t eob_run = 0;
            for (size_t b = 0; b < n; b++) encode_ac_first(&w, coef + b * 64, ss, se, 0, &eob_run, &ht->ac[cls]);
            if (eob_run > 0) flush_eob_run(&w, &eob_run, &ht->ac[cls]);
        }
        bw_flush(&w);
    }
}

/* jpeg/mod.rs:333-373, same order of checks. */
static int validate(const po_options *o, size_t data_len, int check_len)
{
    if (o->quality == 0 || o->quality > 100) return PO_ERR_INVALID_QUALITY;
    if (o->has_restart && o->restart_interval == 0) return PO_ERR_INVALID_RESTART;
    if (o->width == 0 || o->height == 0) return PO_ERR_INVALID_DIMENSIONS;
    if (o->width > 65535 || o->height > 65535) return PO_ERR_IMAGE_TOO_LARGE;
    if (o->color_type != PO_RGB && o->color_type != PO_GRAY) return PO_ERR_UNSUPPORTED_COLOR;
    size_t bpp = o->color_type == PO_RGB ? 3 : 1;
    if (check_len && data_len != (size_t)o->width * o->height * bpp) return PO_ERR_INVALID_DATA_LENGTH;
    return PO_OK;
}

int po_encode_jpeg_from_coeffs(const int16_t *y, const int16_t *cb, const int16_t *cr,
                               const po_options *o, uint8_t **out, size_t *out_len)
{
    int rc = validate(o, 0, 0);
    if (rc) return rc;
    /* (trellis_quant alone: no effect on the baseline path, jpeg/mod.rs:1408-1563 never reads it.
       Progressive WITH optimised tables needs the pixels — the statistics come from the plain
       quantiser, not from this tuple: po_encode_jpeg below.) */
    if (o->progressive && o->optimize_huffman) return PO_ERR_UNSUPPORTED_OPTION;
    uint8_t lzz[64], czz[64];
    float ql[64], qc[64];
    po_quant_tables(o->quality, lzz, czz, ql, qc);
    hufftables ht;
    if (o->optimize_huffman) {
        uint64_t dc[2][12], ac[2][256];
        po_symbol_histograms(y, cb, cr, o, dc, ac);
        optimized_tables(&ht, dc, ac, o->color_type != PO_GRAY);
    } else {
        std_tables(&ht);
    }
    bytevec v = {0};
    bv_reserve(&v, (size_t)o->width * o->height / 4 + 1024);
    put_headers(&v, o, lzz, czz, &ht);
    if (o->progressive) {
        size_t yb, cbn;
        po_coeff_geometry(o->width, o->height, o->color_type, o->subsampling, &yb, &cbn);
        progressive_scans(&v, y, yb, cb, cr, cbn, &ht);
    } else {
        bitw w = {&v, 0, 8};
        walk_ctx c = {0, &w, &ht, NULL, NULL};
        walk_scan(&c, y, cb, cr, o);
        bw_flush(&w);
    }
    bv_be16(&v, 0xFFD9);
    if (v.oom) { free(v.p); return PO_ERR_NOMEM; }
    *out = v.p;
    *out_len = v.n;
    return PO_OK;
}

int po_encode_jpeg(const uint8_t *data, size_t data_len, const po_options *o, uint8_t **out,
                   size_t *out_len)
{
    int rc = validate(o, data_len, 1);
    if (rc) return rc;
    size_t yb, cbn;
    po_coeff_geometry(o->width, o->height, o->color_type, o->subsampling, &yb, &cbn);
    int16_t *y = (int16_t *)malloc((yb + 2 * cbn + 1) * 64 * sizeof(int16_t));
    if (!y) return PO_ERR_NOMEM;
    int16_t *cb = y + yb * 64, *cr = cb + cbn * 64;
    po_jpeg_coeffs(data, o->width, o->height, o->color_type, o->subsampling, o->quality, y, cb, cr, 1);
    if (!o->progressive) {
        rc = po_encode_jpeg_from_coeffs(y, cb, cr, o, out, out_len);
        free(y);
        return rc;
    }
    /* progressive (jpeg/mod.rs:376-419): tables first — optimised ones from the statistics of the
       PLAIN quantiser (build_optimized_huffman_tables :684-824 never uses trellis) — then the
       coefficients again, through the trellis quantiser if asked (compute_all_coefficients :932). */
    uint8_t lzz[64], czz[64];
    float ql[64], qc[64];
    po_quant_tables(o->quality, lzz, czz, ql, qc);
    hufftables ht;
    if (o->optimize_huffman) {
        uint64_t dc[2][12], ac[2][256];
        po_options base = *o;
        base.progressive = 0;
        po_symbol_histograms(y, cb, cr, &base, dc, ac);
        optimized_tables(&ht, dc, ac, o->color_type != PO_GRAY);
    } else {
        std_tables(&ht);
    }
    if (o->trellis_quant)
        po_jpeg_coeffs_ex(data, o->width, o->height, o->color_type, o->subsampling, o->quality, y, cb, cr, 1, 1);
    bytevec v = {0};
    bv_reserve(&v, (size_t)o->width * o->height / 4 + 1024);
    put_headers(&v, o, lzz, czz, &ht);
    progressive_scans(&v, y, yb, cb, cr, cbn, &ht);
    bv_be16(&v, 0xFFD9);
    free(y);
    if (v.oom) { free(v.p); return PO_ERR_NOMEM; }
    *out = v.p;
    *out_len = v.n;
    return PO_OK;
}

int po_encode_jpeg_flat(const uint8_t *data, size_t data_len, uint32_t w, uint32_t h,
                        uint8_t color_type, uint8_t quality, uint8_t preset, int s420,
                        uint8_t **out, size_t *out_len)
{ /* wasm.rs:113-142 + jpeg/mod.rs:162-216 presets */
    if (color_type != PO_GRAY && color_type != PO_RGB) return PO_ERR_UNSUPPORTED_COLOR;
    po_options o;
    memset(&o, 0, sizeof o);
    o.width = w; o.height = h; o.color_type = color_type; o.quality = quality;
    if (preset == 0) { /* fast */
    } else if (preset == 2) { o.optimize_huffman = 1; o.progressive = 1; o.trellis_quant = 1; }
    else { o.optimize_huffman = 1; }
    o.subsampling = s420 ? PO_S420 : PO_S444;
    return po_encode_jpeg(data, data_len, &o, out, out_len);
}

void po_free(void *p) { free(p); }

const char *po_strerror(int code)
{
    switch (code) {
    case PO_OK: return "ok";
    case PO_ERR_INVALID_QUALITY: return "Invalid quality: must be 1-100";
    case PO_ERR_INVALID_RESTART: return "Invalid restart interval: must be 1-65535 (or None to disable)";
    case PO_ERR_INVALID_DIMENSIONS: return "Invalid image dimensions";
    case PO_ERR_IMAGE_TOO_LARGE: return "Image exceeds maximum dimension 65535";
    case PO_ERR_UNSUPPORTED_COLOR: return "Unsupported color type for this format";
    case PO_ERR_INVALID_DATA_LENGTH: return "Invalid pixel data length";
    case PO_ERR_UNSUPPORTED_OPTION: return "option outside the restated path (progressive/trellis)";
    case PO_ERR_NOMEM: return "out of memory";
    default: return "unknown";
    }
}
