// png_filter.hip — gfx950 kernel of the PNG row-filter stage: what the
// reference's apply_filters (src/png/filter.rs:51-206) hands to DEFLATE — one filter-type byte
// plus the filtered row, for every row — and the per-row sums from which the zlib wrapper's
// Adler-32 (src/simd/fallback.rs:8-25, computed once over the whole filtered stream,
// src/compress/deflate.rs:1044) is combined.
//
// Rows are independent on the encode side (every predictor reads ORIGINAL neighbours): one
// workgroup per row.  Pass 1 scores the candidate filters (score_filter: sum of |byte as i8|,
// fallback.rs:93) with four bytes per register — SWAR subtract / average, Paeth in packed 16-bit
// lanes, v_sad_u8 for the score — and reduces the scores across the workgroup; thread-uniform code
// then replays the reference's decision sequence (adaptive_filter :302-393 with its early exits,
// adaptive_filter_fast :474-527, or a fixed filter); pass 2 recomputes only the winning filter,
// stores the row and accumulates the Adler sums.  The row is read twice (second time from L2),
// written once: no intermediate in HBM.  Byte work bounded by HBM and VALU issue; no MFMA.
#include <hip/hip_runtime.h>

#include "jpeg_kernels.hpp"
#include "png_filter.hpp"
#include "png_filter_math.h"

namespace pixo_dev {
namespace {
using namespace pixo_png; // the per-group arithmetic (png_filter_math.h)

constexpr int kThreads = 256;
// One guarded dword (k may be negative or past the end): used by the unaligned variant for
// everything and by the aligned variant for a row's last, partial group.
template <bool FAST> __device__ __forceinline__ uint32_t load_dword(const uint8_t *row, int k, int nbytes)
{
    if (k < 0 || 4 * k >= nbytes) return 0;
    if (FAST) return *reinterpret_cast<const uint32_t *>(row + 4 * k);
    uint32_t v = 0;
#pragma unroll
    for (int b = 0; b < 4; b++)
        if (4 * k + b < nbytes) v |= (uint32_t)row[4 * k + b] << (8 * b);
    return v;
}

// FAST (row base and row length multiples of 4): a group that lies wholly inside the row is read
// with one 16-byte and one 8-byte load per row, no branch near them (the 8 bytes before the
// row's first group do not exist: address clamped, value zeroed by a select).
template <int BPP, bool FAST>
__device__ __forceinline__ void load_six(const uint8_t *row, int k0, int n, bool whole, uint32_t *x)
{
    if (FAST && whole) {
        const uint2 l = *reinterpret_cast<const uint2 *>(row + 4 * (k0 >= 2 ? k0 - 2 : 0));
        const uint4 c = *reinterpret_cast<const uint4 *>(row + 4 * k0);
        x[0] = k0 >= 2 ? l.x : 0u; x[1] = k0 >= 2 ? l.y : 0u; // (k0 is a multiple of 4)
        x[2] = c.x; x[3] = c.y; x[4] = c.z; x[5] = c.w;
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) x[i] = load_dword<FAST>(row, k0 - 2 + i, n);
    }
}

// NEED: bit 0 left neighbours, bit 1 the row above, bit 2 its left neighbours
template <int BPP, bool FAST, int NEED = 7>
__device__ __forceinline__ void load_group(const uint8_t *row, const uint8_t *prev, int k0, int n, Group &g)
{
    uint32_t x[6] = {0, 0, 0, 0, 0, 0}, u[6] = {0, 0, 0, 0, 0, 0};
    const bool whole = 4 * (k0 + 4) <= n;
    if (NEED & 1) load_six<BPP, FAST>(row, k0, n, whole, x);
    else if (FAST && whole) { const uint4 c = *reinterpret_cast<const uint4 *>(row + 4 * k0); x[2] = c.x; x[3] = c.y; x[4] = c.z; x[5] = c.w; }
    else {
#pragma unroll
        for (int i = 2; i < 6; i++) x[i] = load_dword<FAST>(row, k0 - 2 + i, n);
    }
    if (prev && (NEED & 4)) load_six<BPP, FAST>(prev, k0, n, whole, u);
    else if (prev && (NEED & 2)) {
        if (FAST && whole) { const uint4 c = *reinterpret_cast<const uint4 *>(prev + 4 * k0); u[2] = c.x; u[3] = c.y; u[4] = c.z; u[5] = c.w; }
        else {
#pragma unroll
            for (int i = 2; i < 6; i++) u[i] = load_dword<FAST>(prev, k0 - 2 + i, n);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        g.cur[j] = x[2 + j]; g.up[j] = u[2 + j];
        g.left[j] = left_of<BPP>(x, j); g.ul[j] = left_of<BPP>(u, j);
        const int rem = n - 4 * (k0 + j);
        g.valid[j] = rem >= 4 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : (1u << (8 * rem)) - 1u);
    }
}

template <int BPP, bool FAST>
__device__ __forceinline__ void load_raw(const uint8_t *row, const uint8_t *prev, int k0, int n, Raw &r)
{
    const bool whole = 4 * (k0 + 4) <= n;
    load_six<BPP, FAST>(row, k0, n, whole, r.x);
    if (prev) load_six<BPP, FAST>(prev, k0, n, whole, r.u);
    else {
#pragma unroll
        for (int i = 0; i < 6; i++) r.u[i] = 0;
    }
}
__device__ __forceinline__ unsigned long long wg_sum(unsigned long long v, unsigned long long *lds)
{ // 256 threads; every thread gets the total
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    return lds[0] + lds[1] + lds[2] + lds[3];
}

// v + (v of the lane the DPP control names; 0 where that lane does not exist or the row is masked out)
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ uint32_t dpp_add(uint32_t v)
{
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{ // the wavefront's total, uniform (the caller guarantees it fits 32 bits): four DPP additions inside every row of 16
  // lanes, two across the rows, lane 63 read back — 7 instructions where six rounds of __shfl_xor were 6 ds_bpermute + 6
  // additions + their address arithmetic, for each of the 8 totals of a row
    v = dpp_add<0xB1>(v);       // quad_perm:[1,0,3,2]
    v = dpp_add<0x4E>(v);       // quad_perm:[2,3,0,1]
    v = dpp_add<0x141>(v);      // row_half_mirror
    v = dpp_add<0x140>(v);      // row_mirror: every lane of a row holds the row's total
    v = dpp_add<0x142, 0xA>(v); // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v); // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

struct Args {
    const uint8_t *data;
    uint8_t *out;
    unsigned long long *row_sums; // [height][2]: sum of bytes, position-weighted sum (Adler partials)
    const int *forced;            // sequential AdaptiveFast: rows >= 1 use *forced as a fixed filter
    int *winner0;                 //   ... written by row 0
    uint64_t row_bytes;
    uint32_t height, first_row;
    int strategy;
    uint32_t stage_bytes; // dynamic LDS for the staged write-out, 0 = rows too long: direct stores
    uint32_t bitmap_off;  // Bigrams: byte offset of the 8 KiB "pair seen" bitmap inside the dynamic LDS
    uint32_t late_half;   // the second thousand of a launch's workgroups starts late (a whole MI355X holds 2048 at once: launch_png_filter_rows)
};

// Pass 2 for filter F.  The output row starts at byte y * (n + 1) of the stream — a different
// alignment for every row — so the filtered bytes are staged in LDS (filter byte at offset 15, row
// byte i at 16 + i: aligned 16-byte writes) and written out as 16-byte chunks aligned in GLOBAL
// memory: each chunk is five aligned LDS dwords shifted by a row-uniform byte count.  The few
// bytes before the first / after the last aligned chunk are stored singly.  Rows too long for the
// LDS stage (a.stage_bytes == 0) store unaligned dwords directly.
// Filtered dwords of one group -> Adler partial sums + stage (or direct stores for rows too long to stage).
// (T: 64-bit sums in general; 32 bits hold a thread's share of a row of at most 16 KiB)
template <int F, class T>
__device__ __forceinline__ void emit_group(const Group &g, int k0, int n, T L, bool staged, uint8_t *stage,
                                           uint8_t *orow, T &s1, T &s2)
{
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = filtered(F, g, j) & g.valid[j];
    // the group's 16 bytes sit at output positions p = 1 + 4 k0 + i, weight L - p = (L - (4 k0 + 16)) + (15 - i); bytes
    // past the end of the row are zero.  (Unsigned arithmetic: the first factor of a row's last, partial group may
    // "be negative" — the sum is right modulo 2^32 / 2^64 and the true value fits.)
    uint32_t sum, ramp;
    adler_terms16(v, sum, ramp);
    s1 += sum;
    s2 += (L - (T)(4 * k0 + 16)) * sum + ramp;
    if (staged) {
        *reinterpret_cast<uint4 *>(stage + 16 + 4 * k0) = make_uint4(v[0], v[1], v[2], v[3]); // (zero beyond the row)
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint8_t *dst = orow + 1 + 4 * (k0 + j);
            if (g.valid[j] == 0xFFFFFFFFu) __builtin_memcpy(dst, &v[j], 4);
            else for (int b = 0; b < 4; b++) if (4 * (k0 + j) + b < n) dst[b] = (uint8_t)(v[j] >> (8 * b));
        }
    }
}

// The staged row (filter byte at LDS offset 15, row byte i at 16 + i) -> global memory as 16-byte chunks
// aligned in GLOBAL memory: each chunk is five aligned LDS dwords shifted by a row-uniform byte count; the
// few bytes before the first / after the last aligned chunk are stored singly.
template <int NT> __device__ __forceinline__ void flush_stage_body(const uint8_t *stage, uint8_t *orow, int n, int tid)
{
    // stream byte p of this row (0 = filter byte) sits at LDS offset 15 + p
    const int total = n + 1;
    const int head = (int)((16 - (reinterpret_cast<uintptr_t>(orow) & 15)) & 15); // bytes before the first aligned chunk
    const int h = head < total ? head : total;
    const int chunks = (total - h) / 16, tail = total - h - 16 * chunks;
    if (tid < h) orow[tid] = stage[15 + tid];
    if (tid < tail) orow[h + 16 * chunks + tid] = stage[15 + h + 16 * chunks + tid];
    const int t = 15 + h, r = t & 3; // LDS offset of the first chunk; r is uniform
    const uint32_t *w = reinterpret_cast<const uint32_t *>(stage) + (t >> 2);
    for (int c = tid; c < chunks; c += NT) {
        uint32_t d[5];
#pragma unroll
        for (int i = 0; i < 5; i++) d[i] = w[4 * c + i];
        uint4 o;
        o.x = __builtin_amdgcn_alignbyte(d[1], d[0], r); o.y = __builtin_amdgcn_alignbyte(d[2], d[1], r);
        o.z = __builtin_amdgcn_alignbyte(d[3], d[2], r); o.w = __builtin_amdgcn_alignbyte(d[4], d[3], r);
        *reinterpret_cast<uint4 *>(orow + h + 16 * c) = o;
    }
}
template <int NT> __device__ __forceinline__ void flush_stage(const uint8_t *stage, uint8_t *orow, int n, int tid)
{
    __syncthreads();
    flush_stage_body<NT>(stage, orow, n, tid);
}

// Pass 2 for filter F.  The output row starts at byte y * (n + 1) of the stream — a different
// alignment for every row — so the filtered bytes are staged in LDS and written out by flush_stage.
// Rows too long for the LDS stage (a.stage_bytes == 0) store unaligned dwords directly.
template <int BPP, bool FAST, int F, int NEED>
__device__ __forceinline__ void write_row(const Args &a, uint32_t y, const uint8_t *row, const uint8_t *prev, int n,
                                          unsigned long long &s1, unsigned long long &s2)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
    const int ndw = (n + 3) / 4, per_iter = kThreads * 4;
    uint8_t *orow = a.out + (size_t)y * (a.row_bytes + 1);
    const unsigned long long L = (unsigned long long)n + 1; // bytes of the output row; byte p has weight L - p
    const bool staged = a.stage_bytes != 0;
    if (threadIdx.x == 0) {
        if (staged) stage[15] = (uint8_t)F; else orow[0] = (uint8_t)F;
        s1 = (unsigned)F; s2 = L * (unsigned)F;
    }
    for (int k0 = (int)threadIdx.x * 4; k0 < ndw; k0 += per_iter) {
        Group g;
        load_group<BPP, FAST, NEED>(row, prev, k0, n, g);
        emit_group<F, unsigned long long>(g, k0, n, L, staged, stage, orow, s1, s2);
    }
    if (staged) flush_stage<kThreads>(stage, orow, n, (int)threadIdx.x);
}

// The register-resident form (rows of at most kRegIters x 4 KiB with 256 threads, x 8 KiB with 512; always
// staged): the workgroup loaded
// the whole row and the row above ONCE, all loads in flight together; scoring and the winning filter
// both work from those registers.
constexpr int kRegIters = 4;
template <int BPP, int F, int NT, int ITERS>
__device__ __forceinline__ void stage_row_regs(const Args &a, uint32_t y, const Raw *raw, int n, unsigned long long *acc64, int tid)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
    const int ndw = (n + 3) / 4, per_iter = NT * 4; // (256 or 512 threads)
    uint8_t *orow = a.out + (size_t)y * (a.row_bytes + 1);
    const uint32_t L = (uint32_t)n + 1;
    uint32_t s1 = 0, s2 = 0; // this thread's 64 bytes: s2 < 16 x 16385 x 1020 < 2^32
    if (tid == 0) { stage[15] = (uint8_t)F; s1 = (unsigned)F; s2 = L * (unsigned)F; }
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int k0 = tid * 4 + it * per_iter;
        if (k0 < ndw) {
            Group g;
            if (4 * (k0 + 4) <= n) group_of<BPP, false>(raw[it], k0, n, g);
            else group_of<BPP, true>(raw[it], k0, n, g);
            emit_group<F, uint32_t>(g, k0, n, L, true, stage, orow, s1, s2);
        }
    }
    // row totals: 32-bit butterflies inside the wavefront (s2 in two 16-bit halves so that the wave totals
    // fit), one LDS atomic per wavefront; flush_stage's barrier orders them for the reader
    const uint32_t w1 = wave_sum_u32(s1), lo = wave_sum_u32(s2 & 0xFFFFu), hi = wave_sum_u32(s2 >> 16);
    if ((tid & 63) == 0) {
        atomicAdd(&acc64[0], (unsigned long long)w1);
        atomicAdd(&acc64[1], ((unsigned long long)hi << 16) + lo);
    }
}
template <int BPP, int F, int NT, int ITERS>
__device__ __forceinline__ void write_row_regs(const Args &a, uint32_t y, const Raw *raw, int n, unsigned long long *acc64, int tid)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
    stage_row_regs<BPP, F, NT, ITERS>(a, y, raw, n, acc64, tid);
    flush_stage<NT>(stage, a.out + (size_t)y * (a.row_bytes + 1), n, tid);
}

// Bigrams (filter.rs:406-472, score_bigrams :635-649): the score of a candidate is the number of DISTINCT
// adjacent byte pairs of the filtered row.  A 65536-bit "seen" bitmap in LDS, one atomic OR per pair
// (any one-to-one pair -> bit mapping counts the same: the little-endian halfword at each byte
// position), then a population count.  The pair that straddles two 16-byte groups needs the next
// group's first filtered dword: recomputed by the same thread (its loads hit the cache lines the
// neighbouring thread fetches anyway) — no cross-thread exchange, no staging of the row.
template <int BPP, bool FAST, int F, int NEED>
__device__ __forceinline__ unsigned long long bigram_score(const uint8_t *row, const uint8_t *prev, int n, uint32_t *bitmap,
                                                           unsigned long long *red)
{
    for (int i = threadIdx.x; i < 2048; i += kThreads) bitmap[i] = 0;
    __syncthreads();
    const int ndw = (n + 3) / 4, per_iter = kThreads * 4;
    for (int k0 = (int)threadIdx.x * 4; k0 < ndw; k0 += per_iter) {
        Group g;
        load_group<BPP, FAST, NEED>(row, prev, k0, n, g);
        uint32_t v[5];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = filtered(F, g, j);
        v[4] = 0;
        if (k0 + 4 < ndw) {
            Group h;
            load_group<BPP, FAST, NEED>(row, prev, k0 + 4, n, h);
            v[4] = filtered(F, h, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t key[4];
            const int cnt = bigram_keys(v[j], v[j + 1], n - 1 - 4 * (k0 + j), key);
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (i < cnt) atomicOr(&bitmap[key[i] >> 5], 1u << (key[i] & 31u));
        }
    }
    __syncthreads();
    unsigned count = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) count += __builtin_popcount(bitmap[threadIdx.x * 8 + i]);
    return wg_sum(count, red);
}

// Three instantiations by register need: K_GENERAL (fixed filters; adaptive strategies on rows too long to
// hold: two passes over the row, 44 VGPRs), K_REGS (adaptive strategies, the row in registers, ~100),
// K_BIGRAMS (~110).
// K_REGS512: the same with 512 threads, rows of up to 32 KiB.
enum { K_GENERAL = 0, K_REGS = 1, K_BIGRAMS = 2, K_REGS512 = 3 };
template <int BPP, bool FAST, int KIND, int ITERS = kRegIters>
__global__ __launch_bounds__(KIND == K_REGS512 ? 2 * kThreads : kThreads) void png_filter_kernel(const Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
    __shared__ unsigned long long red[20];
    __shared__ unsigned long long acc64[2];
    __shared__ unsigned int acc32[8];
    // Workgroup b runs on XCD b % 8 (round-robin dispatch) and every XCD has its own L2: rows are
    // handed out in chunks of 32 consecutive rows per XCD, so that the row above — the neighbouring
    // workgroup's own row — is found in the same L2 instead of being fetched from HBM a second time
    // (HBM reads 202 -> 92 MB per 4096x4096 RGBA image).  Chunks, not one band per XCD: eight bands
    // 8 MiB apart hit the same HBM channels at the same time and stream 30 % slower.
    const uint32_t nb = gridDim.x, full = nb & ~255u;
    uint32_t y = blockIdx.x;
    // The chip holds 2048 of these workgroups (8 per CU): the first generation of a tall image loads all at once, scores all
    // at once, stores all at once.  Its second half starts one s_sleep (~3.4 us) late: 42.9 -> 40.8 us for the 4096-row image
    // (two sleeps 41.6, four 43.0, graded quarters 41.6-44.5; profiles/r03_png_stagger_ab.txt).
    if (KIND == K_REGS && a.late_half && nb >= 2048u && (y >> 10) == 1u) __builtin_amdgcn_s_sleep(127);
    if (y < full) { const uint32_t xcd = y & 7u, i = y >> 3; y = (((i >> 5) * 8u + xcd) << 5) + (i & 31u); }
    y += a.first_row;
    const int n = (int)a.row_bytes; // < 2^31 (checked by the launcher)
    const uint8_t *row = a.data + (size_t)y * a.row_bytes;
    const uint8_t *prev = y ? row - a.row_bytes : nullptr;
    constexpr int NT = KIND == K_REGS512 ? 2 * kThreads : kThreads;
    const int ndw = (n + 3) / 4, per_iter = NT * 4;
    int strategy = a.forced ? *a.forced : a.strategy;

    int f = strategy;
    if (KIND == K_BIGRAMS) {
        uint32_t *bitmap = reinterpret_cast<uint32_t *>(stage + a.bitmap_off);
        unsigned long long tot[5];
        tot[F_NONE] = bigram_score<BPP, FAST, F_NONE, 0>(row, prev, n, bitmap, red);
        tot[F_SUB] = bigram_score<BPP, FAST, F_SUB, 1>(row, prev, n, bitmap, red);
        tot[F_UP] = bigram_score<BPP, FAST, F_UP, 2>(row, prev, n, bitmap, red);
        tot[F_AVG] = bigram_score<BPP, FAST, F_AVG, 3>(row, prev, n, bitmap, red);
        tot[F_PAETH] = bigram_score<BPP, FAST, F_PAETH, 7>(row, prev, n, bitmap, red);
        f = decide_bigrams(tot);
        __syncthreads(); // the write-out below reuses the dynamic LDS
    } else if (KIND == K_REGS || KIND == K_REGS512) { // (the launcher checked: strategy > Paeth, ndw <= kRegIters * per_iter, staged)
        // the whole row (and the row above) in registers: one round of loads, all in flight together
        if (threadIdx.x < 8) acc32[threadIdx.x] = 0;
        if (threadIdx.x < 2) acc64[threadIdx.x] = 0;
        Raw raw[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int k0 = (int)threadIdx.x * 4 + it * per_iter;
            if (it * per_iter < ndw) load_raw<BPP, FAST>(row, prev, k0, n, raw[it]); // (uniform; k0 >= ndw loads zeros)
            else {
#pragma unroll
                for (int i = 0; i < 6; i++) { raw[it].x[i] = 0; raw[it].u[i] = 0; }
            }
        }
        __syncthreads(); // (the zeroed totals; the loads are needed from here on anyway)
        uint32_t sc[5] = {0, 0, 0, 0, 0};
        const bool fast = strategy == PNG_S_ADAPTIVE_FAST;
        if (fast) { // (uniform)
#pragma unroll
            for (int it = 0; it < ITERS; it++) {
                const int k0 = (int)threadIdx.x * 4 + it * per_iter;
                if (k0 >= ndw) continue;
                if (4 * (k0 + 4) <= n) score_group<BPP, false, true>(raw[it], k0, n, sc);
                else score_group<BPP, true, true>(raw[it], k0, n, sc);
            }
        } else {
#pragma unroll
            for (int it = 0; it < ITERS; it++) {
                const int k0 = (int)threadIdx.x * 4 + it * per_iter;
                if (k0 >= ndw) continue;
                if (4 * (k0 + 4) <= n) score_group<BPP, false, false>(raw[it], k0, n, sc);
                else score_group<BPP, true, false>(raw[it], k0, n, sc);
            }
        }
        // row totals (< 2^32: at most 16 KiB x 128): 32-bit butterflies, one LDS atomic per wavefront and score
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (fast && (i == F_NONE || i == F_AVG)) continue;
            const uint32_t t = wave_sum_u32(sc[i]);
            if ((threadIdx.x & 63) == 0) atomicAdd(&acc32[i], t);
        }
        __syncthreads();
        uint32_t tot[5]; // (uniform, < 2^32: the decision runs on the scalar unit)
#pragma unroll
        for (int i = 0; i < 5; i++) tot[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)acc32[i]);
        f = decide<uint32_t>(strategy, tot, (uint32_t)n);
        if (a.winner0 && y == 0 && threadIdx.x == 0) *a.winner0 = f;
        switch (f) {
        case F_NONE: write_row_regs<BPP, F_NONE, NT, ITERS>(a, y, raw, n, acc64, (int)threadIdx.x); break;
        case F_SUB: write_row_regs<BPP, F_SUB, NT, ITERS>(a, y, raw, n, acc64, (int)threadIdx.x); break;
        case F_UP: write_row_regs<BPP, F_UP, NT, ITERS>(a, y, raw, n, acc64, (int)threadIdx.x); break;
        case F_AVG: write_row_regs<BPP, F_AVG, NT, ITERS>(a, y, raw, n, acc64, (int)threadIdx.x); break;
        default: write_row_regs<BPP, F_PAETH, NT, ITERS>(a, y, raw, n, acc64, (int)threadIdx.x); break;
        }
        if (threadIdx.x == 0) { a.row_sums[2 * (size_t)y] = acc64[0]; a.row_sums[2 * (size_t)y + 1] = acc64[1]; }
        return;
    } else if (strategy > PNG_S_PAETH) {
        // pass 1: scores of the candidates (AdaptiveFast never looks at None / Average)
        uint32_t sc[5] = {0, 0, 0, 0, 0};
        const bool fast = strategy == PNG_S_ADAPTIVE_FAST;
        for (int k0 = (int)threadIdx.x * 4; k0 < ndw; k0 += per_iter) {
            Raw r;
            load_raw<BPP, FAST>(row, prev, k0, n, r);
            if (fast) {
                if (4 * (k0 + 4) <= n) score_group<BPP, false, true>(r, k0, n, sc);
                else score_group<BPP, true, true>(r, k0, n, sc);
            } else {
                if (4 * (k0 + 4) <= n) score_group<BPP, false, false>(r, k0, n, sc);
                else score_group<BPP, true, false>(r, k0, n, sc);
            }
        }
        unsigned long long tot[5];
#pragma unroll
        for (int i = 0; i < 5; i++) tot[i] = wg_sum(sc[i], red);
        f = decide<unsigned long long>(strategy, tot, (unsigned long long)n);
    }
    if (a.winner0 && y == 0 && threadIdx.x == 0) *a.winner0 = f;

    // pass 2: the winning filter -> output row (filter byte + n bytes), Adler partial sums
    unsigned long long s1 = 0, s2 = 0;
    switch (f) { // uniform: one filter's code and loads per row
    case F_NONE: write_row<BPP, FAST, F_NONE, 0>(a, y, row, prev, n, s1, s2); break;
    case F_SUB: write_row<BPP, FAST, F_SUB, 1>(a, y, row, prev, n, s1, s2); break;
    case F_UP: write_row<BPP, FAST, F_UP, 2>(a, y, row, prev, n, s1, s2); break;
    case F_AVG: write_row<BPP, FAST, F_AVG, 3>(a, y, row, prev, n, s1, s2); break;
    default: write_row<BPP, FAST, F_PAETH, 7>(a, y, row, prev, n, s1, s2); break;
    }
    const unsigned long long t1 = wg_sum(s1, red), t2 = wg_sum(s2, red);
    if (threadIdx.x == 0) { a.row_sums[2 * (size_t)y] = t1; a.row_sums[2 * (size_t)y + 1] = t2; }
}

// Bigrams with the row in registers (round 3; rows of at most kRegIters x 4 KiB, staged): the row and the row above are
// loaded ONCE for the five candidates and the write-out (the two-pass form above reads them six times and, for the pair
// that straddles two 16-byte groups, loads and filters the neighbouring group's first dword a second time).  Here that
// dword comes from the neighbouring thread through a 4 KiB exchange array in LDS — thread t, group `it` is entry
// 256 it + t, its successor simply the next entry — and two bitmaps alternate between the candidates, so that a candidate
// costs two barriers: | filter, publish first dwords | A | read successor's, 16 atomic ORs per group | B | population
// count of the bitmap, clear it for the candidate after next |.
template <int BPP, bool FAST, int F>
__device__ __forceinline__ void bigram_candidate(const Raw *raw, int n, int ndw, int tid, uint32_t *bitmap, uint32_t *xch, unsigned int *count)
{
    constexpr int per_iter = kThreads * 4;
    uint32_t v[kRegIters][4];
#pragma unroll
    for (int it = 0; it < kRegIters; it++) {
        const int k0 = tid * 4 + it * per_iter;
        if (k0 >= ndw) continue;
        Group g;
        group_of<BPP, false>(raw[it], k0, n, g);
#pragma unroll
        for (int j = 0; j < 4; j++) v[it][j] = filtered(F, g, j);
        xch[it * kThreads + tid] = v[it][0];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kRegIters; it++) {
        const int k0 = tid * 4 + it * per_iter;
        if (k0 >= ndw) continue;
        const uint32_t next = k0 + 4 < ndw ? xch[it * kThreads + tid + 1] : 0u; // the following group's first filtered dword
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t key[4];
            const int cnt = bigram_keys(v[it][j], j < 3 ? v[it][j + 1] : next, n - 1 - 4 * (k0 + j), key);
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (i < cnt) atomicOr(&bitmap[key[i] >> 5], 1u << (key[i] & 31u));
        }
    }
    __syncthreads();
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (uint32_t)__builtin_popcount(bitmap[tid * 8 + i]); bitmap[tid * 8 + i] = 0; }
    c = wave_sum_u32(c);
    if ((tid & 63) == 0) atomicAdd(count, c);
}

template <int BPP, bool FAST>
__global__ __launch_bounds__(kThreads) void png_bigrams_regs_kernel(const Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
    __shared__ unsigned long long acc64[2];
    __shared__ unsigned int cnt[5];
    const uint32_t nb = gridDim.x, full = nb & ~255u; // rows in chunks of 32 per XCD (png_filter_kernel)
    uint32_t y = blockIdx.x;
    if (y < full) { const uint32_t xcd = y & 7u, i = y >> 3; y = (((i >> 5) * 8u + xcd) << 5) + (i & 31u); }
    y += a.first_row;
    const int n = (int)a.row_bytes, ndw = (n + 3) / 4, tid = (int)threadIdx.x;
    const uint8_t *row = a.data + (size_t)y * a.row_bytes;
    const uint8_t *prev = y ? row - a.row_bytes : nullptr;
    uint32_t *bm0 = reinterpret_cast<uint32_t *>(stage + a.bitmap_off), *bm1 = bm0 + 2048, *xch = bm1 + 2048;
    if (tid < 5) cnt[tid] = 0;
    if (tid < 2) acc64[tid] = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) bm0[i * kThreads + tid] = 0; // both bitmaps
    Raw raw[kRegIters];
#pragma unroll
    for (int it = 0; it < kRegIters; it++) {
        const int k0 = tid * 4 + it * kThreads * 4;
        if (it * kThreads * 4 < ndw) load_raw<BPP, FAST>(row, prev, k0, n, raw[it]);
        else {
#pragma unroll
            for (int i = 0; i < 6; i++) { raw[it].x[i] = 0; raw[it].u[i] = 0; }
        }
    }
    __syncthreads();
    bigram_candidate<BPP, FAST, F_NONE>(raw, n, ndw, tid, bm0, xch, &cnt[F_NONE]);
    bigram_candidate<BPP, FAST, F_SUB>(raw, n, ndw, tid, bm1, xch, &cnt[F_SUB]);
    bigram_candidate<BPP, FAST, F_UP>(raw, n, ndw, tid, bm0, xch, &cnt[F_UP]);
    bigram_candidate<BPP, FAST, F_AVG>(raw, n, ndw, tid, bm1, xch, &cnt[F_AVG]);
    bigram_candidate<BPP, FAST, F_PAETH>(raw, n, ndw, tid, bm0, xch, &cnt[F_PAETH]);
    __syncthreads();
    unsigned long long tot[5];
#pragma unroll
    for (int i = 0; i < 5; i++) tot[i] = cnt[i];
    const int f = decide_bigrams(tot);
    switch (f) {
    case F_NONE: write_row_regs<BPP, F_NONE, kThreads, kRegIters>(a, y, raw, n, acc64, tid); break;
    case F_SUB: write_row_regs<BPP, F_SUB, kThreads, kRegIters>(a, y, raw, n, acc64, tid); break;
    case F_UP: write_row_regs<BPP, F_UP, kThreads, kRegIters>(a, y, raw, n, acc64, tid); break;
    case F_AVG: write_row_regs<BPP, F_AVG, kThreads, kRegIters>(a, y, raw, n, acc64, tid); break;
    default: write_row_regs<BPP, F_PAETH, kThreads, kRegIters>(a, y, raw, n, acc64, tid); break;
    }
    if (tid == 0) { a.row_sums[2 * (size_t)y] = acc64[0]; a.row_sums[2 * (size_t)y + 1] = acc64[1]; }
}

template <int BPP> hipError_t launch_bpp(const Args &a, uint32_t rows, bool fast, hipStream_t s)
{
    const uint64_t ndw = (a.row_bytes + 3) / 4;
    if (a.strategy == PNG_S_BIGRAMS && a.stage_bytes != 0 && ndw <= (uint64_t)kRegIters * kThreads * 4) {
        const uint32_t lds = a.stage_bytes + 2 * 8192u + 4u * (kRegIters * kThreads + 1); // stage | two bitmaps | exchange array
        if (fast) hipLaunchKernelGGL((png_bigrams_regs_kernel<BPP, true>), dim3(rows), dim3(kThreads), lds, s, a);
        else hipLaunchKernelGGL((png_bigrams_regs_kernel<BPP, false>), dim3(rows), dim3(kThreads), lds, s, a);
    } else if (a.strategy == PNG_S_BIGRAMS) {
        const uint32_t lds = a.stage_bytes + 8192u;
        if (fast) hipLaunchKernelGGL((png_filter_kernel<BPP, true, K_BIGRAMS>), dim3(rows), dim3(kThreads), lds, s, a);
        else hipLaunchKernelGGL((png_filter_kernel<BPP, false, K_BIGRAMS>), dim3(rows), dim3(kThreads), lds, s, a);
    } else if (a.strategy > PNG_S_PAETH && !a.forced && ndw <= (uint64_t)kRegIters * 2 * kThreads * 4 && a.stage_bytes != 0) {
        // rows of up to 16 KiB: 256 threads hold them; up to 32 KiB: 512 threads
        if (ndw <= (uint64_t)kRegIters * kThreads * 4) {
            if (fast) hipLaunchKernelGGL((png_filter_kernel<BPP, true, K_REGS>), dim3(rows), dim3(kThreads), a.stage_bytes, s, a);
            else hipLaunchKernelGGL((png_filter_kernel<BPP, false, K_REGS>), dim3(rows), dim3(kThreads), a.stage_bytes, s, a);
        } else if (fast) hipLaunchKernelGGL((png_filter_kernel<BPP, true, K_REGS512>), dim3(rows), dim3(2 * kThreads), a.stage_bytes, s, a);
        else hipLaunchKernelGGL((png_filter_kernel<BPP, false, K_REGS512>), dim3(rows), dim3(2 * kThreads), a.stage_bytes, s, a);
    } else if (fast) hipLaunchKernelGGL((png_filter_kernel<BPP, true, K_GENERAL>), dim3(rows), dim3(kThreads), a.stage_bytes, s, a);
    else hipLaunchKernelGGL((png_filter_kernel<BPP, false, K_GENERAL>), dim3(rows), dim3(kThreads), a.stage_bytes, s, a);
    return hipGetLastError();
}

hipError_t launch_rows(const Args &a, uint32_t rows, uint32_t bpp, bool fast, hipStream_t s)
{
    switch (bpp) {
    case 1: return launch_bpp<1>(a, rows, fast, s);
    case 2: return launch_bpp<2>(a, rows, fast, s);
    case 3: return launch_bpp<3>(a, rows, fast, s);
    case 4: return launch_bpp<4>(a, rows, fast, s);
    case 6: return launch_bpp<6>(a, rows, fast, s);
    case 8: return launch_bpp<8>(a, rows, fast, s);
    default: return hipErrorInvalidValue;
    }
}
} // namespace

hipError_t launch_png_filter(const void *d_data, uint32_t width, uint32_t height, uint32_t bpp, int strategy,
                             bool sequential_fast, void *d_out, unsigned long long *d_row_sums, int *d_scratch,
                             hipStream_t stream)
{
    return launch_png_filter_rows(d_data, width, height, bpp, strategy, sequential_fast, d_out, d_row_sums, d_scratch, 0, height, stream);
}

// Rows [first_row, first_row + rows) of the image (the row above the first one must be in d_data already): the host-pixel
// entry uploads, filters and downloads a large image band by band.  The stateful AdaptiveFast needs the whole image.
hipError_t launch_png_filter_rows(const void *d_data, uint32_t width, uint32_t height, uint32_t bpp, int strategy,
                                  bool sequential_fast, void *d_out, unsigned long long *d_row_sums, int *d_scratch,
                                  uint32_t first_row, uint32_t rows, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    if (first_row + rows > height) return hipErrorInvalidValue;
    Args a;
    a.data = static_cast<const uint8_t *>(d_data);
    a.out = static_cast<uint8_t *>(d_out);
    a.row_sums = d_row_sums;
    a.row_bytes = (uint64_t)width * bpp;
    if (a.row_bytes >= 0x7FFFFFF0ull) return hipErrorInvalidValue; // 32-bit byte indices inside a row
    a.height = height;
    a.strategy = strategy;
    // LDS stage: 16 bytes in front (filter byte at 15), the row rounded up to whole 16-byte groups of
    // 4 KiB iterations, 16 bytes of slack for the fifth dword of the last chunk
    const uint64_t stage = 16 + ((a.row_bytes + 15) & ~15ull) + 32;
    a.stage_bytes = stage <= 48 * 1024 ? (uint32_t)stage : 0u;
    a.bitmap_off = a.stage_bytes;
    a.forced = nullptr; a.winner0 = nullptr; a.first_row = 0;
    a.late_half = (rows >= 2048u && whole_chip_device()) ? 1u : 0u;
    const bool fast = reinterpret_cast<uintptr_t>(d_data) % 4 == 0 && a.row_bytes % 4 == 0;
    if (strategy == PNG_S_ADAPTIVE_FAST && sequential_fast && height > 1) {
        // sequential AdaptiveFast (filter.rs:147-167): the first row's winner (always Sub, Up or
        // Paeth) is forced on every later row; two launches, no host round trip
        if (first_row != 0 || rows != height) return hipErrorInvalidValue;
        a.winner0 = d_scratch;
        hipError_t e = launch_rows(a, 1, bpp, fast, stream);
        if (e != hipSuccess) return e;
        a.winner0 = nullptr; a.forced = d_scratch; a.first_row = 1;
        return launch_rows(a, height - 1, bpp, fast, stream);
    }
    a.first_row = first_row;
    return launch_rows(a, rows, bpp, fast, stream);
}

} // namespace pixo_dev
