// jpeg_trellis.hip — trellis quantisation of raw DCT blocks on gfx950: the quantiser
// the reference's progressive path uses when `trellis_quant` is set (quantize_dct, src/jpeg/mod.rs:
// 968-976 -> src/jpeg/trellis.rs).  One lane = one block; the Viterbi state of a lane (a few dozen
// floats, 63 x 8 back-pointers) lives in scratch memory.  Divergent, latency-bound integer/float
// control code — orders of magnitude below the coefficient kernel's rate and still ~100x one CPU core.
#include <hip/hip_runtime.h>

#include "jpeg_trellis.h"
#include "jpeg_trellis.hpp"

#pragma clang fp contract(off)

namespace pixo_dev {
namespace {
__global__ __launch_bounds__(64) void trellis_kernel(const float *raw, const float *q, float prescale, int16_t *out, uint64_t nblocks)
{
    const uint64_t b = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= nblocks) return;
    float dct[64], qq[64];
    const float4 *src = reinterpret_cast<const float4 *>(raw + b * 64);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float4 v = src[i];
        dct[4 * i] = v.x * prescale; dct[4 * i + 1] = v.y * prescale; dct[4 * i + 2] = v.z * prescale; dct[4 * i + 3] = v.w * prescale;
    }
    for (int i = 0; i < 64; i++) qq[i] = q[i];
    int16_t res[64];
    uint32_t trail[63 * 8];
    uint8_t counts[63];
    pixo_trellis::quantize_block(dct, qq, res, trail, counts);
    uint32_t *dst = reinterpret_cast<uint32_t *>(out + b * 64);
    for (int i = 0; i < 32; i++) dst[i] = (uint32_t)(uint16_t)res[2 * i] | ((uint32_t)(uint16_t)res[2 * i + 1] << 16);
}
} // namespace

hipError_t launch_trellis(const float *d_raw, const float *d_q, float prescale, int16_t *d_out, uint64_t nblocks, hipStream_t s)
{
    if (nblocks == 0) return hipSuccess;
    hipLaunchKernelGGL(trellis_kernel, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, s, d_raw, d_q, prescale, d_out, nblocks);
    return hipGetLastError();
}
} // namespace pixo_dev
