// jpeg_kernels.hip — gfx950 kernels of the JPEG pixel pipeline and their launcher.
//
// The per-tile body lives in jpeg_tile.h (shared with the CPU emulation harness in
// tests/emu).  This file adds the persistent __global__ loop, the tile -> image mapping
// and the host-side launch function.  Written for CDNA4 only: 64-lane wavefronts,
// 256-thread workgroups of four role-specialised wavefronts (one producer, three consumers),
// ~50-58 KiB LDS per workgroup (two resident per CU = two busy wavefronts per SIMD), a grid
// of (resident workgroups per CU) x 256 CUs that walks the tiles with a stride.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "jpeg_kernels.hpp"
#include "jpeg_tile.h"

#pragma clang fp contract(off)

namespace pixo_dev {
using namespace pixo_tile;

struct KArgs {
    const uint8_t *px;
    int16_t *y, *cb, *cr;
    const float *qt;
    uint32_t W, H, units_x, units_y, tiles_x, tiles_y, batch, fast;
    size_t px_stride; // bytes between consecutive images of a batch
    size_t y_stride;  // i16 elements between images
    size_t c_stride;
    unsigned long long *dbg; // PIXO_TIMING builds only: per-workgroup cycle sums
};

// Workgroup barrier that orders LDS only.  __syncthreads() also drains vmcnt, which would
// make every barrier wait for the next tile's prefetch loads and the previous tile's stores.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// A tile is (image, tile column, tile row): three SGPRs.  The full TileCtx is rebuilt from the
// kernel arguments where it is needed instead of being carried (three live copies of it cost
// ~30 SGPRs and pushed the kernel to the 102-SGPR limit).
struct TileId {
    uint32_t img, tx, ty;
};

__device__ __forceinline__ TileId locate(const KArgs &a, uint32_t t)
{
    const uint32_t per_img = a.tiles_x * a.tiles_y;
    TileId id;
    id.img = t / per_img;
    const uint32_t r = t - id.img * per_img;
    id.ty = r / a.tiles_x;
    id.tx = r - id.ty * a.tiles_x;
    return id;
}

__device__ __forceinline__ TileCtx ctx_of(const KArgs &a, uint32_t img)
{
    TileCtx c;
    c.px = a.px + (size_t)img * a.px_stride;
    c.y = a.y + (size_t)img * a.y_stride;
    c.cb = a.cb ? a.cb + (size_t)img * a.c_stride : nullptr;
    c.cr = a.cr ? a.cr + (size_t)img * a.c_stride : nullptr;
    c.qt = a.qt;
    c.W = a.W; c.H = a.H; c.units_x = a.units_x; c.units_y = a.units_y; c.fast = a.fast;
    return c;
}

// Role-specialised persistent workgroup.  Tiles t = blockIdx.x + i * gridDim.x, i = 0..n-1.
//   producer (wave 3):   for i: fill planar[i & 1] with tile i; barrier_i
//   consumers (0..2):    for i: barrier_i; transform + store tile i from planar[i & 1]
// Both sides execute exactly n barriers.  The producer refills planar[i & 1] (tile i + 2) only
// after barrier_{i+1}, which the consumers reach only after they are done with tile i, so one
// barrier per tile is the whole protocol.  While the consumers work on tile i the producer
// converts tile i + 1 and has tile i + 2's loads in flight: each item's registers are reloaded
// for the next tile as soon as the item has been converted, so HBM reads are spread evenly
// over the iteration and never waited for at the point of issue.
template <int MODE, bool FAST>
__global__ __launch_bounds__(kThreads, 2) void jpeg_coeffs_kernel(const KArgs a)
{
    typedef Geo<MODE> G;
    __shared__ __attribute__((aligned(16))) uint8_t lds[lds_bytes<MODE>()];
    uint8_t *const stage = lds + 2 * G::planar;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t total = a.tiles_x * a.tiles_y * a.batch;
    const uint32_t first = blockIdx.x, stride = gridDim.x;
    if (first >= total) return;

    if (wave == 3) {
        // Two register sets: while set X (tile t, loaded one iteration ago) is converted, set Y
        // receives tile t + stride.  The explicit vmcnt(0) sits BEFORE the new loads are issued,
        // so it only ever waits for loads that have had a whole iteration to land; the loads are
        // unconditional (the last iteration re-reads its own tile) so that no branch joins —
        // and therefore no compiler-inserted wait — follow them.
        uint32_t ra[G::items * G::item_regs], rb[G::items * G::item_regs];
        TileId id = locate(a, first);
        TileCtx c = ctx_of(a, id.img);
#pragma unroll
        for (int k = 0; k < G::items; k++) producer_load_item<MODE, FAST>(c, id.tx, id.ty, k, lane, &ra[k * G::item_regs]);
        uint32_t buf = 0, t = first;
#ifdef PIXO_TIMING
        unsigned long long tm[5] = {0, 0, 0, 0, 0}, acc[4] = {0, 0, 0, 0};
#define PIXO_T(i) tm[i] = __builtin_readcyclecounter(); if (i > 0) acc[i - 1] += tm[i] - tm[i - 1];
#else
#define PIXO_T(i)
#endif
        for (;;) {
#define PIXO_PRODUCER_HALF(X, Y)                                                                   \
            {                                                                                      \
                const uint32_t tn = t + stride;                                                    \
                const bool more = tn < total;                                                      \
                const TileCtx cc = c;                                                              \
                const uint32_t ctx = id.tx;                                                        \
                id = locate(a, more ? tn : t);                                                     \
                c = ctx_of(a, id.img);                                                             \
                PIXO_T(0) __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0): set X has landed */     \
                PIXO_T(1) _Pragma("unroll") for (int k = 0; k < G::items; k++)                     \
                    producer_load_item<MODE, FAST>(c, id.tx, id.ty, k, lane, &Y[k * G::item_regs]); \
                uint8_t *planar = lds + buf * G::planar;                                           \
                PIXO_T(2) _Pragma("unroll") for (int k = 0; k < G::items; k++) {                   \
                    PIXO_PRODUCER_CONVERT(X)                                                       \
                }                                                                                  \
                buf ^= 1;                                                                          \
                PIXO_T(3) lds_barrier(); /* planar buffer of tile t handed over */                 \
                PIXO_T(4)                                                                          \
                if (!more) break;                                                                  \
                t = tn;                                                                            \
            }
#if !defined(PIXO_ABLATE) || PIXO_ABLATE < 2 // (timing experiments: >=2 drops the colour conversion)
#define PIXO_PRODUCER_CONVERT(X)                                                  \
    producer_fix_item<MODE, FAST>(cc, ctx, k, lane, &X[k * G::item_regs]);        \
    producer_color_item<MODE>(k, lane, &X[k * G::item_regs], planar);          \
    __builtin_amdgcn_sched_barrier(0); /* one item at a time: ~40 temporaries, not 16 x 40 */
#else
#define PIXO_PRODUCER_CONVERT(X) \
    for (int i = 0; i < G::item_regs; i++) asm volatile("" ::"v"(X[k * G::item_regs + i]));
#endif
            PIXO_PRODUCER_HALF(ra, rb)
            PIXO_PRODUCER_HALF(rb, ra)
#undef PIXO_PRODUCER_HALF
#undef PIXO_PRODUCER_CONVERT
        }
#ifdef PIXO_TIMING
        if (lane == 0 && a.dbg) for (int i = 0; i < 4; i++) a.dbg[blockIdx.x * 16 + i] = acc[i];
#endif
    } else {
        uint32_t buf = 0;
#ifdef PIXO_TIMING
        unsigned long long tm[5] = {0, 0, 0, 0, 0}, acc[4] = {0, 0, 0, 0};
#endif
        for (uint32_t t = first; t < total; t += stride) {
            PIXO_T(0) lds_barrier(); // tile t's planar buffer is complete
            PIXO_T(1) const TileId id = locate(a, t);
            const uint8_t *planar = lds + buf * G::planar;
            float v[64];
#if !defined(PIXO_ABLATE) || PIXO_ABLATE < 1 // (>=1 drops the transform)
            consumer_rows<MODE>(wave, lane, planar, v);
            PIXO_T(2) consumer_cols_quant<MODE>(wave, lane, a.qt, v, stage);
#endif
            PIXO_T(3) consumer_store<MODE>(ctx_of(a, id.img), id.tx, id.ty, wave, lane, stage);
            PIXO_T(4) buf ^= 1;
        }
#ifdef PIXO_TIMING
        if (lane == 0 && a.dbg) for (int i = 0; i < 4; i++) a.dbg[blockIdx.x * 16 + 4 + wave * 4 + i] = acc[i];
#endif
    }
}

// Resident workgroups per CU, computed from the kernel's own resource usage: gfx950 has 160 KiB
// of LDS and 512 VGPRs per lane per SIMD (allocation granule 8), and one wavefront of each
// 4-wave workgroup lands on each SIMD.  (hipOccupancyMaxActiveBlocksPerMultiprocessor answers 1
// for the 57 KiB-LDS kernels — it still budgets 64 KiB of LDS per CU — which would halve the
// grid and leave every wavefront alone on its SIMD, at half the VALU issue rate.)
template <int MODE, bool FAST> static int blocks_per_cu()
{
    static int cached = 0;
    if (!cached) {
        hipFuncAttributes attr;
        int n = 1;
        if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(jpeg_coeffs_kernel<MODE, FAST>)) == hipSuccess) {
            const int regs = ((attr.numRegs > 0 ? attr.numRegs : 256) + 7) / 8 * 8;
            const int by_vgpr = 512 / regs;
            const int by_lds = (160 * 1024) / (attr.sharedSizeBytes > 0 ? (int)attr.sharedSizeBytes : lds_bytes<MODE>());
            n = by_vgpr < by_lds ? by_vgpr : by_lds;
            if (getenv("PIXO_HIP_DEBUG"))
                fprintf(stderr, "pixo_hip: mode %d fast %d numRegs %d sharedSizeBytes %zu maxDyn %d -> %d workgroups/CU\n", MODE,
                        (int)FAST, attr.numRegs, attr.sharedSizeBytes, attr.maxDynamicSharedSizeBytes, n);
        } else if (getenv("PIXO_HIP_DEBUG")) {
            fprintf(stderr, "pixo_hip: hipFuncGetAttributes failed\n");
        }
        if (const char *e = getenv("PIXO_HIP_BLOCKS_PER_CU")) n = atoi(e); // tuning experiments
        cached = n < 1 ? 1 : (n > 8 ? 8 : n);
    }
    return cached;
}

template <int MODE, bool FAST> static hipError_t launch_mode(KArgs &a, hipStream_t s)
{
    a.tiles_x = (a.units_x + Geo<MODE>::units_x - 1) / Geo<MODE>::units_x;
    a.tiles_y = (a.units_y * (MODE == M420 ? 16u : 8u) + Geo<MODE>::tile_h - 1) / Geo<MODE>::tile_h;
    const uint64_t total64 = (uint64_t)a.tiles_x * a.tiles_y * a.batch;
    if (total64 > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const uint32_t total = (uint32_t)total64;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t resident = (uint32_t)blocks_per_cu<MODE, FAST>() * (uint32_t)cus;
    // equal number of tiles per workgroup: rounds = ceil(total / resident), grid = ceil(total / rounds)
    const uint32_t rounds = (total + resident - 1) / resident;
    const uint32_t grid = (total + rounds - 1) / rounds;
    hipLaunchKernelGGL((jpeg_coeffs_kernel<MODE, FAST>), dim3(grid), dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_jpeg_coeffs(const void *d_px, uint32_t W, uint32_t H, bool gray, bool s420,
                              uint32_t batch, void *d_y, void *d_cb, void *d_cr,
                              const float *d_qt, hipStream_t stream)
{
    KArgs a;
    a.px = static_cast<const uint8_t *>(d_px);
    a.y = static_cast<int16_t *>(d_y);
    a.cb = static_cast<int16_t *>(d_cb);
    a.cr = static_cast<int16_t *>(d_cr);
    a.qt = d_qt;
    a.dbg = nullptr;
#ifdef PIXO_TIMING
    if (const char *e = getenv("PIXO_DBG_PTR")) a.dbg = reinterpret_cast<unsigned long long *>(strtoull(e, nullptr, 0));
#endif
    a.W = W; a.H = H; a.batch = batch;
    const uint32_t unit = (!gray && s420) ? 16 : 8;
    a.units_x = (W + unit - 1) / unit;
    a.units_y = (H + unit - 1) / unit;
    const uint32_t bpp = gray ? 1 : 3;
    const size_t row_bytes = static_cast<size_t>(W) * bpp;
    a.px_stride = row_bytes * H;
    // dword loads need 4-byte aligned rows in every image of the batch
    a.fast = (reinterpret_cast<uintptr_t>(d_px) % 4 == 0) && (row_bytes % 4 == 0) &&
             (batch == 1 || a.px_stride % 4 == 0);
    const size_t units = static_cast<size_t>(a.units_x) * a.units_y;
    a.y_stride = (unit == 16 ? 4 * units : units) * 64;
    a.c_stride = units * 64;
    // FAST additionally needs at least one whole 4-pixel group per row (address clamp W - 4)
    const bool fast = a.fast && W >= 4;
    if (gray) return fast ? launch_mode<MGRAY, true>(a, stream) : launch_mode<MGRAY, false>(a, stream);
    if (s420) return fast ? launch_mode<M420, true>(a, stream) : launch_mode<M420, false>(a, stream);
    return fast ? launch_mode<M444, true>(a, stream) : launch_mode<M444, false>(a, stream);
}

} // namespace pixo_dev
