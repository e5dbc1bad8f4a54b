/* pixo_int_oracle.c — TEST INFRASTRUCTURE.  CPU restatement of the reference's INTEGER DCT family
 * (SURVEY.md §8 row a17), each function citing the lines it follows (leerob/pixo v0.4.1):
 *
 *   po_dct_2d_integer        src/jpeg/dct.rs:61-186   (fix_mul :32-34: (i64 a*b) >> 13, truncating PER PRODUCT;
 *                                                      pass 1 scales only outputs 0 and 4 by << 2; pass 2 (x + 16) >> 5)
 *   po_dct_2d_fast           src/jpeg/dct.rs:535-568  (constant block: out[0] = 8 * value, rest 0; else dct_2d_integer —
 *                                                      the scalar result is what dct_2d_avx2 must equal, x86_64.rs:1053)
 *   po_quantize_block_integer src/jpeg/dct.rs:570-583 ((c +- q/2) / q, truncating toward zero)
 *   po_rgb_to_ycbcr_2p16     src/simd/x86_64.rs:1330-1420 (2^16 coefficients 19595/38470/7471, -11056/-21712/32768,
 *                                                      32768/-27440/-5328, + 32768, arithmetic >> 16; Y - 128, Cb, Cr centred)
 *
 * PARITY UNPINNED beyond the reference's own unit-test values: `encode()` never calls this family, so
 * it is stripped from the reference's wasm build and no runnable reference exists for it.  The restatement is pinned
 * on every exact value the reference's unit tests state (dct.rs:867-1183, x86_64.rs:2077-2250:
 * tests/test_integer_mode.py) and otherwise rests on the source text.
 *
 * po_jpeg_coeffs_integer composes them the way the product's labelled secondary mode does (there is no composition in
 * the reference to follow): per 8x8 block with the edge replication of extract_block (src/jpeg/mod.rs:1565-1606),
 * 4:4:4 and gray only, quantiser tables = `luminance_table_int` / `chrominance_table_int` (quantize.rs:56-78).
 */
#include <stdint.h>
#include <string.h>

#include "pixo_oracle.h"

#define CONST_BITS 13
#define PASS1_BITS 2
static int32_t fix_mul(int32_t a, int32_t b) { return (int32_t)(((int64_t)a * (int64_t)b) >> CONST_BITS); } /* dct.rs:32-34 */

enum {
    FIX_0_298631336 = 2446, FIX_0_390180644 = 3196, FIX_0_541196100 = 4433, FIX_0_765366865 = 6270,
    FIX_0_899976223 = 7373, FIX_1_175875602 = 9633, FIX_1_501321110 = 12299, FIX_1_847759065 = 15137,
    FIX_1_961570560 = 16069, FIX_2_053119869 = 16819, FIX_2_562915447 = 20995, FIX_3_072711026 = 25172
}; /* dct.rs:38-49 */

/* one 1-D pass over 8 values spaced `stride` apart; pass 1 (final = 0): dct.rs:66-124, pass 2 (final = 1): :128-183 */
static void pass(const int32_t *in, int32_t *out, int stride, int final)
{
    const int32_t d0 = in[0], d1 = in[stride], d2 = in[2 * stride], d3 = in[3 * stride];
    const int32_t d4 = in[4 * stride], d5 = in[5 * stride], d6 = in[6 * stride], d7 = in[7 * stride];
    int32_t tmp0 = d0 + d7, tmp1 = d1 + d6, tmp2 = d2 + d5, tmp3 = d3 + d4;
    int32_t tmp10 = tmp0 + tmp3, tmp12 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp13 = tmp1 - tmp2;
    tmp0 = d0 - d7; tmp1 = d1 - d6; tmp2 = d2 - d5; tmp3 = d3 - d4;
    const int descale = PASS1_BITS + 3, half = 1 << (descale - 1);
    int32_t z1 = fix_mul(tmp12 + tmp13, FIX_0_541196100);
    int32_t o0, o2, o4, o6;
    if (!final) {
        o0 = (tmp10 + tmp11) << PASS1_BITS;
        o4 = (tmp10 - tmp11) << PASS1_BITS;
        o2 = z1 + fix_mul(tmp12, FIX_0_765366865);
        o6 = z1 - fix_mul(tmp13, FIX_1_847759065);
    } else {
        o0 = (tmp10 + tmp11 + half) >> descale;
        o4 = (tmp10 - tmp11 + half) >> descale;
        o2 = (z1 + fix_mul(tmp12, FIX_0_765366865) + half) >> descale;
        o6 = (z1 - fix_mul(tmp13, FIX_1_847759065) + half) >> descale;
    }
    tmp10 = tmp0 + tmp3; tmp11 = tmp1 + tmp2; tmp12 = tmp0 + tmp2; tmp13 = tmp1 + tmp3;
    z1 = fix_mul(tmp12 + tmp13, FIX_1_175875602);
    tmp0 = fix_mul(tmp0, FIX_1_501321110);
    tmp1 = fix_mul(tmp1, FIX_3_072711026);
    tmp2 = fix_mul(tmp2, FIX_2_053119869);
    tmp3 = fix_mul(tmp3, FIX_0_298631336);
    tmp10 = fix_mul(tmp10, -FIX_0_899976223);
    tmp11 = fix_mul(tmp11, -FIX_2_562915447);
    tmp12 = fix_mul(tmp12, -FIX_0_390180644) + z1;
    tmp13 = fix_mul(tmp13, -FIX_1_961570560) + z1;
    int32_t o1 = tmp0 + tmp10 + tmp12, o3 = tmp1 + tmp11 + tmp13, o5 = tmp2 + tmp11 + tmp12, o7 = tmp3 + tmp10 + tmp13;
    if (final) { o1 = (o1 + half) >> descale; o3 = (o3 + half) >> descale; o5 = (o5 + half) >> descale; o7 = (o7 + half) >> descale; }
    out[0] = o0; out[stride] = o1; out[2 * stride] = o2; out[3 * stride] = o3;
    out[4 * stride] = o4; out[5 * stride] = o5; out[6 * stride] = o6; out[7 * stride] = o7;
}

void po_dct_2d_integer(const int16_t block[64], int32_t out[64])
{
    int32_t in[64], ws[64];
    for (int i = 0; i < 64; i++) in[i] = block[i];
    for (int r = 0; r < 8; r++) pass(in + 8 * r, ws + 8 * r, 1, 0);
    for (int c = 0; c < 8; c++) pass(ws + c, out + c, 8, 1);
}

void po_dct_2d_fast(const int16_t block[64], int32_t out[64])
{ /* dct.rs:535-551: the constant-block shortcut; everything else goes to the integer transform */
    int constant = 1;
    for (int i = 1; i < 64; i++) constant &= block[i] == block[0];
    if (constant) {
        memset(out, 0, 64 * sizeof(int32_t));
        out[0] = (int32_t)block[0] * 8;
        return;
    }
    po_dct_2d_integer(block, out);
}

void po_quantize_block_integer(const int32_t dct[64], const uint16_t q[64], int16_t out[64])
{ /* dct.rs:570-583; `/` truncates toward zero in Rust and in C */
    for (int i = 0; i < 64; i++) {
        const int32_t qi = q[i], c = dct[i];
        out[i] = (int16_t)(c >= 0 ? (c + (qi >> 1)) / qi : (c - (qi >> 1)) / qi);
    }
}

void po_rgb_to_ycbcr_2p16(uint8_t r8, uint8_t g8, uint8_t b8, int32_t out[3])
{ /* x86_64.rs:1406-1412 (the scalar tail states the same arithmetic as the vector body :1371-1398) */
    const int32_t r = r8, g = g8, b = b8;
    out[0] = ((19595 * r + 38470 * g + 7471 * b + 32768) >> 16) - 128;
    out[1] = (-11056 * r - 21712 * g + 32768 * b + 32768) >> 16;
    out[2] = (32768 * r - 27440 * g - 5328 * b + 32768) >> 16;
}

void po_quant_tables_int(uint8_t quality, uint16_t lum[64], uint16_t chr[64])
{ /* quantize.rs:56-78: the same integers as the f32 tables, natural order */
    uint8_t lzz[64], czz[64];
    float fl[64], fc[64];
    po_quant_tables(quality, lzz, czz, fl, fc);
    for (int i = 0; i < 64; i++) { lum[i] = (uint16_t)fl[i]; chr[i] = (uint16_t)fc[i]; }
}

int po_jpeg_coeffs_integer(const uint8_t *px, uint32_t w, uint32_t h, uint8_t color_type, uint8_t quality, int16_t *y,
                           int16_t *cb, int16_t *cr)
{
    if (color_type != 0 && color_type != 2) return PO_ERR_UNSUPPORTED_COLOR;
    uint16_t ql[64], qc[64];
    po_quant_tables_int(quality, ql, qc);
    const uint32_t bw = (w + 7) / 8, bh = (h + 7) / 8;
    for (uint32_t by = 0; by < bh; by++)
        for (uint32_t bx = 0; bx < bw; bx++) {
            int16_t s[3][64];
            for (int dy = 0; dy < 8; dy++)
                for (int dx = 0; dx < 8; dx++) { /* extract_block: x = min(bx*8+dx, w-1), y likewise */
                    uint32_t x = bx * 8 + dx, yy = by * 8 + dy;
                    if (x > w - 1) x = w - 1;
                    if (yy > h - 1) yy = h - 1;
                    if (color_type == 0) {
                        s[0][dy * 8 + dx] = (int16_t)((int)px[(size_t)yy * w + x] - 128);
                    } else {
                        const uint8_t *p = px + ((size_t)yy * w + x) * 3;
                        int32_t c[3];
                        po_rgb_to_ycbcr_2p16(p[0], p[1], p[2], c);
                        s[0][dy * 8 + dx] = (int16_t)c[0]; s[1][dy * 8 + dx] = (int16_t)c[1]; s[2][dy * 8 + dx] = (int16_t)c[2];
                    }
                }
            const size_t blk = (size_t)by * bw + bx;
            int32_t d[64];
            po_dct_2d_fast(s[0], d);
            po_quantize_block_integer(d, ql, y + blk * 64);
            if (color_type == 2) {
                po_dct_2d_fast(s[1], d);
                po_quantize_block_integer(d, qc, cb + blk * 64);
                po_dct_2d_fast(s[2], d);
                po_quantize_block_integer(d, qc, cr + blk * 64);
            }
        }
    return 0;
}
