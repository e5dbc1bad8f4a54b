// jpeg_int_math.h — per-block arithmetic of the INTEGER secondary mode: the reference's fixed-point
// transform family, which its `encode()` never calls (dead code upstream) but which the north star names.
// Never the default: byte-identical files need the f32 path (jpeg_tile.h); this mode is reached only through
// pixo_hip_jpeg_coeffs_integer[_device].
//
//   dct_2d_integer           src/jpeg/dct.rs:61-186    13-bit fixed point, fix_mul = (i64 a*b) >> 13 truncating per
//                                                      product (:32-34), pass 1: outputs 0 and 4 << 2, pass 2: (x + 16) >> 5
//   dct_2d_fast              src/jpeg/dct.rs:535-568   constant block -> out[0] = 8 * value
//   quantize_block_integer   src/jpeg/dct.rs:570-583   (c +- q/2) / q toward zero
//   rgb_to_ycbcr_row_avx2    src/simd/x86_64.rs:1330-1420   2^16-scaled coefficients, + 32768, >> 16
//
// One lane = one 8x8 block, everything in registers, 32-bit integer VALU (the 64-bit product of fix_mul is
// v_mul_hi_i32 + v_mul_lo_u32 + v_alignbit).  Shared with tests/emu (-DPIXO_EMU): plain integer code.
#pragma once
#include <stdint.h>

#if defined(PIXO_EMU)
#define PIXO_IDEV static inline
#else
#define PIXO_IDEV __device__ __forceinline__
#endif

namespace pixo_int {

PIXO_IDEV int32_t fix_mul(int32_t a, int32_t b) { return (int32_t)(((int64_t)a * (int64_t)b) >> 13); }

// one 1-D pass in place over v[0], v[S], ..., v[7 S]; FINAL: the column pass with its rounding shift
template <int S, bool FINAL> PIXO_IDEV void pass8(int32_t *v)
{
    const int32_t d0 = v[0], d1 = v[S], d2 = v[2 * S], d3 = v[3 * S], d4 = v[4 * S], d5 = v[5 * S], d6 = v[6 * S], d7 = v[7 * S];
    int32_t t0 = d0 + d7, t1 = d1 + d6, t2 = d2 + d5, t3 = d3 + d4;
    int32_t t10 = t0 + t3, t12 = t0 - t3, t11 = t1 + t2, t13 = t1 - t2;
    t0 = d0 - d7; t1 = d1 - d6; t2 = d2 - d5; t3 = d3 - d4;
    constexpr int kDescale = 2 + 3, kHalf = 1 << (kDescale - 1);
    int32_t z1 = fix_mul(t12 + t13, 4433);
    int32_t o0, o2, o4, o6;
    if (!FINAL) {
        o0 = (t10 + t11) << 2; o4 = (t10 - t11) << 2;
        o2 = z1 + fix_mul(t12, 6270); o6 = z1 - fix_mul(t13, 15137);
    } else {
        o0 = (t10 + t11 + kHalf) >> kDescale; o4 = (t10 - t11 + kHalf) >> kDescale;
        o2 = (z1 + fix_mul(t12, 6270) + kHalf) >> kDescale; o6 = (z1 - fix_mul(t13, 15137) + kHalf) >> kDescale;
    }
    t10 = t0 + t3; t11 = t1 + t2; t12 = t0 + t2; t13 = t1 + t3;
    z1 = fix_mul(t12 + t13, 9633);
    t0 = fix_mul(t0, 12299); t1 = fix_mul(t1, 25172); t2 = fix_mul(t2, 16819); t3 = fix_mul(t3, 2446);
    t10 = fix_mul(t10, -7373); t11 = fix_mul(t11, -20995);
    t12 = fix_mul(t12, -3196) + z1; t13 = fix_mul(t13, -16069) + z1;
    int32_t o1 = t0 + t10 + t12, o3 = t1 + t11 + t13, o5 = t2 + t11 + t12, o7 = t3 + t10 + t13;
    if (FINAL) { o1 = (o1 + kHalf) >> kDescale; o3 = (o3 + kHalf) >> kDescale; o5 = (o5 + kHalf) >> kDescale; o7 = (o7 + kHalf) >> kDescale; }
    v[0] = o0; v[S] = o1; v[2 * S] = o2; v[3 * S] = o3; v[4 * S] = o4; v[5 * S] = o5; v[6 * S] = o6; v[7 * S] = o7;
}

// dct_2d_fast: v = 64 level-shifted samples (row-major) -> 64 coefficients, in place
PIXO_IDEV void dct_2d_fast(int32_t *v)
{
    bool constant = true;
#pragma unroll
    for (int i = 1; i < 64; i++) constant = constant && v[i] == v[0];
    if (constant) { // dct.rs:538-551
        const int32_t dc = v[0] * 8;
#pragma unroll
        for (int i = 0; i < 64; i++) v[i] = 0;
        v[0] = dc;
        return;
    }
#pragma unroll
    for (int r = 0; r < 8; r++) pass8<1, false>(v + 8 * r);
#pragma unroll
    for (int c = 0; c < 8; c++) pass8<8, true>(v + c);
}

PIXO_IDEV int32_t quantize_integer(int32_t c, int32_t q) { return c >= 0 ? (c + (q >> 1)) / q : (c - (q >> 1)) / q; }

struct YCbCr { int32_t y, cb, cr; }; // Y level-shifted, Cb / Cr centred
PIXO_IDEV YCbCr rgb_to_ycbcr_2p16(int32_t r, int32_t g, int32_t b)
{
    YCbCr o;
    o.y = ((19595 * r + 38470 * g + 7471 * b + 32768) >> 16) - 128;
    o.cb = (-11056 * r - 21712 * g + 32768 * b + 32768) >> 16;
    o.cr = (32768 * r - 27440 * g - 5328 * b + 32768) >> 16;
    return o;
}

} // namespace pixo_int
