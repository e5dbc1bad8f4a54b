// jpeg_kernels.hip — gfx950 kernels of the JPEG pixel pipeline and their launcher.
//
// The per-tile body lives in jpeg_tile.h (shared with the CPU emulation harness in
// tests/emu).  This file adds the persistent __global__ loop, the tile -> image mapping
// and the host-side launch function.  Written for CDNA4 only: 64-lane wavefronts,
// 256-thread workgroups (4 waves, one per SIMD), 24-32 KiB LDS per workgroup, a grid of
// (resident workgroups per CU) x 256 CUs that walks the tiles with a stride, each
// workgroup prefetching its next tile's pixels into registers while it transforms the
// current one, so HBM reads, VALU work and HBM writes of neighbouring tiles overlap.
#include <hip/hip_runtime.h>

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

// Software pipeline, one call site per phase.  Iteration i of a workgroup (tile i = its i-th
// tile; planar and stage are disjoint LDS areas):
//   1. colour-convert tile i (registers -> LDS planar)       first use of the loaded pixels
//   2. write tile i-1's stage out to HBM, THEN issue tile i+1's global loads
//   3. transform tile i (LDS planar -> LDS stage)             pure VALU/LDS, no vmcnt wait
// Every VMEM operation of an iteration is issued in step 2 and not waited for until step 1 of
// the next iteration, so stores and loads drain under the whole of step 3.  (vmcnt retires in
// order: with the loads older than the stores, a wait for the loads would also be a wait for
// the stores just issued — the previous schedule serialised store drain and VALU work.)
template <int MODE>
__global__ __launch_bounds__(kThreads) void jpeg_coeffs_kernel(const KArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[Geo<MODE>::lds];
    const int tid = threadIdx.x;
    const uint32_t total = a.tiles_x * a.tiles_y * a.batch;
    uint32_t t = blockIdx.x;
    if (t >= total) return;
    Lane<MODE> L;
    TileId prev = locate(a, t), cur = prev, nxt = prev;
    load_tile<MODE>(ctx_of(a, cur.img), cur.tx, cur.ty, tid, L);
    bool have_prev = false;
    for (;;) {
#if !defined(PIXO_ABLATE) || PIXO_ABLATE < 2 // (timing experiments: >=2 drops the colour conversion)
        phase_color<MODE>(tid, L, lds);
#else
        for (int i = 0; i < Geo<MODE>::in_regs; i++) asm volatile("" ::"v"(L.in[i]));
#endif
        lds_barrier(); // planar(cur) complete; stage(prev) was completed before the last barrier
        if (have_prev) phase_store<MODE>(ctx_of(a, prev.img), prev.tx, prev.ty, tid, lds);
        const uint32_t tn = t + gridDim.x;
        const bool more = tn < total;
        if (more) {
            nxt = locate(a, tn);
            load_tile<MODE>(ctx_of(a, nxt.img), nxt.tx, nxt.ty, tid, L);
        }
        lds_barrier(); // stage(prev) read out: phase B may overwrite it
#if !defined(PIXO_ABLATE) || PIXO_ABLATE < 1 // (>=1 drops the transform)
        phase_dct_quant<MODE>(tid, a.qt, lds);
#endif
        prev = cur;
        have_prev = true;
        if (!more) break;
        cur = nxt;
        t = tn;
        lds_barrier(); // planar(cur) consumed: the next colour conversion may overwrite it
    }
    lds_barrier();
    phase_store<MODE>(ctx_of(a, prev.img), prev.tx, prev.ty, tid, lds);
}

// resident workgroups per CU, per mode (queried once per process)
template <int MODE> static int blocks_per_cu()
{
    static int cached = 0;
    if (!cached) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, jpeg_coeffs_kernel<MODE>, kThreads, 0) != hipSuccess || n < 1)
            n = 2;
        cached = n > 8 ? 8 : n;
    }
    return cached;
}

template <int MODE> static hipError_t launch_mode(KArgs &a, hipStream_t s)
{
    a.tiles_x = (a.units_x + Geo<MODE>::units_x - 1) / Geo<MODE>::units_x;
    a.tiles_y = (a.units_y * (MODE == M420 ? 16u : 8u) + Geo<MODE>::tile_h - 1) / Geo<MODE>::tile_h;
    const uint64_t total64 = (uint64_t)a.tiles_x * a.tiles_y * a.batch;
    if (total64 > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const uint32_t total = (uint32_t)total64;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t resident = (uint32_t)blocks_per_cu<MODE>() * (uint32_t)cus;
    // equal number of tiles per workgroup: rounds = ceil(total / resident), grid = ceil(total / rounds)
    const uint32_t rounds = (total + resident - 1) / resident;
    const uint32_t grid = (total + rounds - 1) / rounds;
    hipLaunchKernelGGL(jpeg_coeffs_kernel<MODE>, dim3(grid), dim3(kThreads), 0, s, a);
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
    if (gray) return launch_mode<MGRAY>(a, stream);
    if (s420) return launch_mode<M420>(a, stream);
    return launch_mode<M444>(a, stream);
}

} // namespace pixo_dev
