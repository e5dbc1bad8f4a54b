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

#include <atomic>

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

// ---- the same search on EIGHT LANES per block (round 5; small images) ------------------------------------------------------
// One lane per block makes a wavefront's 63 steps a chain of ~440 vector instructions each: 117 us whatever the image's size,
// and an image of up to 65,536 blocks does not even fill the chip's SIMDs with such wavefronts.  Here a block's eight SURVIVORS
// sit on eight consecutive lanes (a wavefront = eight blocks) and one step is, per lane:
//   * the three non-zero candidates against ITS parent (three costs), then the first strict minimum over the eight parents by
//     three DPP exchanges (quad swaps and the half-row mirror) of 64-bit keys (cost bits, parent) — quantize_block_fast's own
//     v_min_f64 order;
//   * ITS parent's zero successor (the "first of the run-0 group" rule from a ballot over the group's lanes);
//   * the reference's stable sort of the eleven entries as a RANK COUNT: the group's eight zero keys pass through LDS, every
//     lane counts the keys below its own (and, lanes 0..2, below one candidate's) and drops them at their ranks; lane r then
//     picks up the survivor of rank r.  Keys are quantize_block_fast's (cost bits, slot, run, kind, parent): all distinct, so
//     the ranks are the sorting network's positions.
// Back-pointers: a byte per survivor and step in LDS (4 KiB per wavefront); the walk back is a serial chase of 63 LDS bytes,
// the values are then recomputed eight positions a lane.  Same arithmetic, same order, same ties as jpeg_trellis.h — the two
// kernels are held to each other on random images at every quality (tests/test_gpu_progressive.py) and both to the oracle.
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <int CTRL> __device__ __forceinline__ uint64_t dpp64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}
// the smallest key of the group's eight lanes, in all of them
__device__ __forceinline__ uint64_t group_min(uint64_t k)
{
    k = pixo_trellis::min_key(k, dpp64<0xB1>(k));  // quad_perm [1, 0, 3, 2]
    k = pixo_trellis::min_key(k, dpp64<0x4E>(k));  // quad_perm [2, 3, 0, 1]
    k = pixo_trellis::min_key(k, dpp64<0x141>(k)); // row_half_mirror: lane i of eight <- lane 7 - i (the other quad)
    return k;
}
constexpr int kGroupsPerWave = 8;
__global__ __launch_bounds__(64) void trellis_lanes_kernel(const float *raw, const float *q_luma, const float *q_chroma, int16_t *out,
                                                           uint64_t nblocks, uint64_t nluma)
{
    using namespace pixo_trellis;
    __shared__ __attribute__((aligned(16))) uint64_t s_z[kGroupsPerWave][8];       // the zero successors' keys, by parent
    __shared__ __attribute__((aligned(16))) uint64_t s_sorted[kGroupsPerWave][12]; // the eleven keys by rank
    __shared__ uint8_t s_trail[kGroupsPerWave][64][8];
    __shared__ __attribute__((aligned(16))) v4u s_pre[kGroupsPerWave][64]; // per position: the three candidates' distortions, the zero candidate's
    __shared__ uint32_t s_meta[kGroupsPerWave][64];                         // ... and per candidate a byte: size << 4 | kind << 1 | valid
    __shared__ uint8_t s_kind[kGroupsPerWave][64];
    __shared__ __attribute__((aligned(16))) int16_t s_res[kGroupsPerWave][64];
    __shared__ float s_step[128];
    __shared__ float s_bits[256];
    const int lane = threadIdx.x, g = lane >> 3, s = lane & 7;
    const uint64_t first = (uint64_t)blockIdx.x * kGroupsPerWave;
    const uint64_t have = nblocks - first < (uint64_t)kGroupsPerWave ? nblocks - first : (uint64_t)kGroupsPerWave;
    s_step[lane] = q_luma[lane];
    s_step[64 + lane] = q_chroma[lane];
#pragma unroll
    for (int i = 0; i < 4; i++) s_bits[i * 64 + lane] = rate_value(i * 64 + lane);
    __syncthreads();
    const uint64_t B = first + ((uint64_t)g < have ? (uint64_t)g : have - 1); // (a short last wavefront searches its last block again)
    const float *col = raw + (B >> 6) * 4096 + (B & 63); // coefficient i (natural order) of the block: col[64 i]
    const float *steps = s_step + (B < nluma ? 0 : 64);
    // What a step needs of its coefficient does not depend on the search: the candidates' distortions, size categories, kinds and
    // valid bits, and the zero candidate's distortion.  All 63 positions are worked out FIRST, eight a lane, side by side (the
    // divide, the three roundings, the saturating conversions: half of a step's instructions) — the serial loop below only reads them.
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int zz = s + 8 * i, nat = c_zigzag_nat[zz];
        const LanePre p = lanes_prepare(col[nat * 64], steps[nat]);
        s_pre[g][zz] = v4u{f2u(p.dist[0]), f2u(p.dist[1]), f2u(p.dist[2]), f2u(p.dist0)};
        s_meta[g][zz] = p.meta;
    }
    __syncthreads();
    uint32_t cc = s == 0 ? 0u : kNoState, run6 = 0; // this lane's survivor: cost bits, run << 6
    for (int zz = 1; zz < 64; zz++) {
        const v4u pre = s_pre[g][zz];
        LanePre p;
        p.dist[0] = u2f(pre.x); p.dist[1] = u2f(pre.y); p.dist[2] = u2f(pre.z); p.dist0 = u2f(pre.w);
        p.meta = s_meta[g][zz];
        uint64_t cand[3];
#pragma unroll
        for (int j = 0; j < 3; j++) cand[j] = lanes_candidate(group_min(lanes_cost_key(cc, run6, p, j, s, s_bits)), p, j);
        const uint64_t r0 = __builtin_amdgcn_ballot_w64(lanes_alive_run0(cc, run6));
        const uint32_t in_front = (uint32_t)(r0 >> (8 * g)) & ((1u << s) - 1u); // run-0 parents on the group's lower lanes
        const uint64_t zkey = lanes_zero_key(cc, run6, p, s, in_front != 0);
        s_z[g][s] = zkey;
        __syncthreads();
        uint64_t z[8];
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
            const v4u q = *reinterpret_cast<const v4u *>(&s_z[g][t]);
            z[t] = ((uint64_t)q.y << 32) | q.x; z[t + 1] = ((uint64_t)q.w << 32) | q.z;
        }
        const uint64_t mine = s == 0 ? cand[0] : (s == 1 ? cand[1] : cand[2]); // (lanes 0..2 place one candidate each)
        uint32_t rank_z, rank_c;
        lanes_ranks(z, cand, zkey, mine, &rank_z, &rank_c);
        s_sorted[g][rank_z] = zkey;
        s_sorted[g][s < 3 ? rank_c : 11u] = mine; // (slot 11: nobody's)
        __syncthreads();
        uint8_t back;
        lanes_take(s_sorted[g][s], &cc, &run6, &back);
        s_trail[g][zz - 1][s] = back;
    }
    int idx = (int)(uint32_t)group_min(lanes_final_key(cc, run6, s));
    __syncthreads();
    for (int zz = 63; zz >= 1; zz--) { // (every lane of the group walks the same chain)
        const uint32_t f = s_trail[g][zz - 1][idx] & 63u;
        if (s == 0) s_kind[g][zz] = (uint8_t)(f >> 3);
        idx = (int)(f & 7u);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; i++) { // the values, eight positions a lane
        const int zz = s + 8 * i, nat = c_zigzag_nat[zz];
        s_res[g][nat] = (int16_t)lanes_value(col[nat * 64] / steps[nat], zz ? (int)s_kind[g][zz] : 0, zz == 0);
    }
    __syncthreads();
    { // a block's 128 bytes by its eight lanes
        const uint64_t blk = first + (uint64_t)g;
        if ((uint64_t)g < have) *reinterpret_cast<v4u *>(out + blk * 64 + s * 8) = *reinterpret_cast<const v4u *>(&s_res[g][s * 8]);
    }
}
} // namespace

size_t trellis_scratch_bytes(uint64_t nblocks) { return (size_t)((nblocks + 63) / 64) * 63 * 64 * 8; }

static std::atomic<int> g_trellis_form{0}; // (measurements and tests: PIXO_HIP_DEBUG trellis_form=lane|group; may be flipped while other threads launch)
void set_trellis_form(int form) { g_trellis_form.store(form, std::memory_order_relaxed); }
hipError_t launch_trellis(const float *d_raw, const float *d_q_luma, const float *d_q_chroma, int16_t *d_out, uint64_t nblocks,
                          uint64_t nluma, void *d_scratch, hipStream_t s)
{
    if (nblocks == 0) return hipSuccess;
    // eight lanes per block while the one-lane form's wavefronts (64 blocks each) would leave SIMDs idle anyway
    const int form = g_trellis_form.load(std::memory_order_relaxed);
    const bool lanes = form == 2 || (form == 0 && nblocks <= kTrellisLanesBlocks);
    if (lanes)
        hipLaunchKernelGGL(trellis_lanes_kernel, dim3((unsigned)((nblocks + kGroupsPerWave - 1) / kGroupsPerWave)), dim3(64), 0, s, d_raw, d_q_luma,
                           d_q_chroma, d_out, nblocks, nluma);
    else
        hipLaunchKernelGGL(trellis_kernel, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, s, d_raw, d_q_luma, d_q_chroma, d_out, nblocks,
                           nluma, static_cast<uint64_t *>(d_scratch));
    return hipGetLastError();
}
} // namespace pixo_dev
