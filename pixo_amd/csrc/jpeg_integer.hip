// jpeg_integer.hip — the INTEGER secondary mode of the coefficient stage (jpeg_int_math.h has the
// arithmetic and its reference citations).  A labelled side door, not the product's encode path: one lane per 8x8
// block, pixels gathered with the reference's clamp-replicate addressing (extract_block, src/jpeg/mod.rs:1565-1606),
// colour by the 2^16 row formulas, fixed-point transform, integer quantiser, natural-order i16[64] blocks in the
// YCbCrCoefficients layout of 4:4:4 / gray.  Integer work bounded by its ~1,000 VALU operations per block; no MFMA, no
// LDS (every lane's 64 values live in its registers).
#include <hip/hip_runtime.h>

#include "jpeg_int_math.h"
#include "jpeg_kernels.hpp"

namespace pixo_dev {
namespace {
struct IntArgs {
    const uint8_t *px;
    int16_t *y, *cb, *cr;
    uint32_t W, H, bw, bh;
    uint32_t ql[64], qc[64]; // luminance_table_int / chrominance_table_int (natural order), by value: scalar loads
};

__device__ __forceinline__ void store_block(int16_t *dst, const int32_t *v, const uint32_t *q)
{
    uint32_t out[32];
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const int32_t a = pixo_int::quantize_integer(v[2 * i], (int32_t)q[2 * i]);
        const int32_t b = pixo_int::quantize_integer(v[2 * i + 1], (int32_t)q[2 * i + 1]);
        out[i] = ((uint32_t)a & 0xFFFFu) | ((uint32_t)b << 16);
    }
    uint4 *p = reinterpret_cast<uint4 *>(dst);
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = make_uint4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
}

template <bool GRAY> __global__ __launch_bounds__(64) void jpeg_integer_kernel(const IntArgs a)
{
    const uint64_t blk = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (blk >= (uint64_t)a.bw * a.bh) return;
    const uint32_t by = (uint32_t)(blk / a.bw), bx = (uint32_t)(blk - (uint64_t)by * a.bw);
    int32_t vy[64], vcb[GRAY ? 1 : 64], vcr[GRAY ? 1 : 64];
#pragma unroll
    for (int dy = 0; dy < 8; dy++) {
        const uint32_t yy = by * 8 + dy < a.H ? by * 8 + dy : a.H - 1;
        const uint8_t *row = a.px + (size_t)yy * a.W * (GRAY ? 1 : 3);
#pragma unroll
        for (int dx = 0; dx < 8; dx++) {
            const uint32_t x = bx * 8 + dx < a.W ? bx * 8 + dx : a.W - 1;
            if (GRAY) {
                vy[dy * 8 + dx] = (int32_t)row[x] - 128;
            } else {
                const pixo_int::YCbCr c = pixo_int::rgb_to_ycbcr_2p16(row[3 * x], row[3 * x + 1], row[3 * x + 2]);
                vy[dy * 8 + dx] = c.y; vcb[dy * 8 + dx] = c.cb; vcr[dy * 8 + dx] = c.cr;
            }
        }
    }
    pixo_int::dct_2d_fast(vy);
    store_block(a.y + blk * 64, vy, a.ql);
    if (!GRAY) {
        pixo_int::dct_2d_fast(vcb);
        store_block(a.cb + blk * 64, vcb, a.qc);
        pixo_int::dct_2d_fast(vcr);
        store_block(a.cr + blk * 64, vcr, a.qc);
    }
}
} // namespace

hipError_t launch_jpeg_coeffs_integer(const void *d_px, uint32_t W, uint32_t H, bool gray, const uint16_t ql[64], const uint16_t qc[64],
                                      void *d_y, void *d_cb, void *d_cr, hipStream_t stream)
{
    IntArgs a;
    a.px = static_cast<const uint8_t *>(d_px);
    a.y = static_cast<int16_t *>(d_y); a.cb = static_cast<int16_t *>(d_cb); a.cr = static_cast<int16_t *>(d_cr);
    a.W = W; a.H = H; a.bw = (W + 7) / 8; a.bh = (H + 7) / 8;
    for (int i = 0; i < 64; i++) { a.ql[i] = ql[i]; a.qc[i] = qc[i]; }
    const uint64_t blocks = (uint64_t)a.bw * a.bh, groups = (blocks + 63) / 64;
    if (groups > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (gray) hipLaunchKernelGGL(jpeg_integer_kernel<true>, dim3((unsigned)groups), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(jpeg_integer_kernel<false>, dim3((unsigned)groups), dim3(64), 0, stream, a);
    return hipGetLastError();
}

} // namespace pixo_dev
