// jpeg_trellis.hip — trellis quantisation of raw DCT blocks on gfx950: the quantiser
// the reference's progressive path uses when `trellis_quant` is set (quantize_dct, src/jpeg/mod.rs:
// 968-976 -> src/jpeg/trellis.rs).
//
// One lane = one block, one wave = one workgroup = 64 consecutive blocks.  The Viterbi state of a lane
// (8 costs, 8 runs, 12 successor keys) stays in registers (pixo_trellis::quantize_block_fast).  The coefficient
// kernel's raw mode writes the transform already in this kernel's order — [wave][coefficient][lane] — so every step of
// the search is one coalesced load, fetched a step ahead; the code-length estimates are a 256-entry LDS table; the
// back-pointers (8 bytes per lane and coefficient) go to a global scratch laid out [wave][position][lane] so that both
// directions are full-line accesses; the results are collected as i16 in LDS (8 KiB) and leave as whole 128-byte blocks.
// Round 1 staged the wave's 64 x 64 f32 coefficients in LDS: 18 KiB per wavefront = 8 wavefronts per CU, two per SIMD,
// for a kernel whose every step is a chain of dependent LDS look-ups; 10 KiB and four per SIMD now.
// ALU work: ~700 VALU instructions per coefficient position.
#include <hip/hip_runtime.h>

#include "jpeg_trellis.h"
#include "jpeg_trellis.hpp"

#pragma clang fp contract(off)

namespace pixo_dev {
namespace {
constexpr int kOutPitch = 66; // i16 slots between two lanes' result rows in LDS (33 words: no bank conflicts)

__constant__ int c_zigzag_nat[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                     12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                     58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct LaneEnv {
    const float *col;     // this lane's column of the wave's coefficients: natural index i at col[64 i]
    const float *steps;   // this lane's 64 quantiser steps in LDS (natural order)
    const float *table;   // ac_rate for the 256 (run, size) symbols in LDS
    uint64_t *trail;      // this lane's column of the wave's back-pointer scratch
    int16_t *mine;        // this lane's 64 result slots in LDS (natural order)
    __device__ __forceinline__ float coef(int zz) const { return col[c_zigzag_nat[zz] * 64]; }
    __device__ __forceinline__ float step(int zz) const { return steps[c_zigzag_nat[zz]]; }
    __device__ __forceinline__ float rate_at(uint32_t byte_off) const { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(table) + byte_off); }
    __device__ __forceinline__ void trail_put(int pos, uint64_t w) { __builtin_nontemporal_store(w, trail + pos * 64); }
    __device__ __forceinline__ uint64_t trail_get(int pos) const { return __builtin_nontemporal_load(trail + pos * 64); }
    __device__ __forceinline__ void out(int zz, int16_t v) { mine[c_zigzag_nat[zz]] = v; }
};

// Blocks [0, nluma) use q_luma, the rest q_chroma: the three planes of a tuple are one run of blocks, one launch.
__global__ __launch_bounds__(64) void trellis_kernel(const float *raw, const float *q_luma, const float *q_chroma, int16_t *out,
                                                     uint64_t nblocks, uint64_t nluma, uint64_t *trail)
{
    __shared__ __attribute__((aligned(4))) int16_t s_out[64 * kOutPitch];
    __shared__ float s_step[128];
    __shared__ float s_bits[256];
    const int lane = threadIdx.x;
    const uint64_t first = (uint64_t)blockIdx.x * 64;
    const uint64_t have = nblocks - first < 64 ? nblocks - first : 64; // blocks of this wave
    s_step[lane] = q_luma[lane];
    s_step[64 + lane] = q_chroma[lane];
#pragma unroll
    for (int i = 0; i < 4; i++) s_bits[i * 64 + lane] = pixo_trellis::rate_value(i * 64 + lane);
    __syncthreads();
    // (a short last wave searches its last block again in the idle lanes)
    const int mine = lane < (int)have ? lane : (int)have - 1;
    LaneEnv env{raw + (uint64_t)blockIdx.x * 4096 + mine, s_step + (first + mine < nluma ? 0 : 64), s_bits,
                trail + (uint64_t)blockIdx.x * 63 * 64 + lane, s_out + lane * kOutPitch};
    pixo_trellis::quantize_block_fast(env);
    __syncthreads();
    uint32_t *dst = reinterpret_cast<uint32_t *>(out + first * 64);
#pragma unroll
    for (int it = 0; it < 32; it++) {
        const int i = it * 64 + lane, blk = i >> 5, pair = i & 31;
        if (blk < (int)have) __builtin_nontemporal_store(*reinterpret_cast<const uint32_t *>(s_out + blk * kOutPitch + pair * 2), dst + i);
    }
}
} // namespace

size_t trellis_scratch_bytes(uint64_t nblocks) { return (size_t)((nblocks + 63) / 64) * 63 * 64 * 8; }

hipError_t launch_trellis(const float *d_raw, const float *d_q_luma, const float *d_q_chroma, int16_t *d_out, uint64_t nblocks,
                          uint64_t nluma, void *d_scratch, hipStream_t s)
{
    if (nblocks == 0) return hipSuccess;
    hipLaunchKernelGGL(trellis_kernel, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, s, d_raw, d_q_luma, d_q_chroma, d_out, nblocks,
                       nluma, static_cast<uint64_t *>(d_scratch));
    return hipGetLastError();
}
} // namespace pixo_dev
