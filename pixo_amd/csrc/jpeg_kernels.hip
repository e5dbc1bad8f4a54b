// jpeg_kernels.hip — gfx950 kernels of the JPEG pixel pipeline and their launcher.
//
// The per-tile body lives in jpeg_tile.h (shared with the CPU emulation harness in
// tests/emu).  This file adds the __global__ wrappers, the blockIdx -> tile mapping and
// the host-side launch function.  Written for CDNA4 only: 64-lane wavefronts, 256-thread
// workgroups (4 waves, one per SIMD), <= 32 KiB LDS per workgroup so that 5 workgroups
// share a CU's 160 KiB, >= 2048 workgroups per 4096x4096 image (8 per CU).
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
    uint32_t W, H, units_x, units_y, tiles_x, fast;
    size_t px_stride;  // bytes between consecutive images of a batch
    size_t y_stride;   // i16 elements between images
    size_t c_stride;
};

template <int MODE>
__global__ __launch_bounds__(kThreads) void jpeg_coeffs_kernel(const KArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[lds_bytes<MODE>()];
    const int tid = threadIdx.x;
    const uint32_t tile_y = blockIdx.x / a.tiles_x;
    const uint32_t tile_x = blockIdx.x - tile_y * a.tiles_x;
    const size_t img = blockIdx.y;

    TileCtx c;
    c.px = a.px + img * a.px_stride;
    c.y = a.y + img * a.y_stride;
    c.cb = a.cb ? a.cb + img * a.c_stride : nullptr;
    c.cr = a.cr ? a.cr + img * a.c_stride : nullptr;
    c.qt = a.qt;
    c.W = a.W; c.H = a.H; c.units_x = a.units_x; c.units_y = a.units_y; c.fast = a.fast;

    Lane<MODE> L;
    if (tile_is_interior<MODE>(c, tile_x, tile_y))
        phase_load<MODE, true>(c, tile_x, tile_y, tid, L);
    else
        phase_load<MODE, false>(c, tile_x, tile_y, tid, L);
    phase_color<MODE>(tid, L, lds);
    __syncthreads();
    const int cls = phase_fetch<MODE>(tid, lds, L);
    __syncthreads(); // planar samples are in registers; the stage may now overwrite them
    phase_dct_quant<MODE>(tid, cls, c.qt, L, lds);
    __syncthreads();
    phase_store<MODE>(c, tile_x, tile_y, tid, lds);
}

template <int MODE> static hipError_t launch_mode(const KArgs &a, uint32_t batch, hipStream_t s)
{
    const uint32_t tiles_y = (a.units_y * (MODE == M420 ? 16u : 8u) + Geo<MODE>::tile_h - 1) / Geo<MODE>::tile_h;
    dim3 grid(a.tiles_x * tiles_y, batch, 1);
    hipLaunchKernelGGL(jpeg_coeffs_kernel<MODE>, grid, dim3(kThreads), 0, s, a);
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
    a.W = W; a.H = H;
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
    if (gray) {
        a.tiles_x = (a.units_x + Geo<MGRAY>::units_x - 1) / Geo<MGRAY>::units_x;
        return launch_mode<MGRAY>(a, batch, stream);
    }
    if (s420) {
        a.tiles_x = (a.units_x + Geo<M420>::units_x - 1) / Geo<M420>::units_x;
        return launch_mode<M420>(a, batch, stream);
    }
    a.tiles_x = (a.units_x + Geo<M444>::units_x - 1) / Geo<M444>::units_x;
    return launch_mode<M444>(a, batch, stream);
}

} // namespace pixo_dev
