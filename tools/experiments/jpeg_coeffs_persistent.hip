// jpeg_coeffs_persistent.hip — EXPERIMENT, a stand-alone binary, not part of the library:
// the coefficient kernel as PERSISTENT workgroups that take several 512x16-pixel tiles each, the next tile's RGB bytes
// landing in LDS by LDS-DMA (global_load_lds_dwordx4, no registers involved) while phase B of the current tile runs —
// against the product's one-tile-per-workgroup kernel (through the C ABI of libpixo_hip.so), same image, same timing
// method as bench.py (HIP events over back-to-back launches, 7 rotating buffer sets).  Same arithmetic: the per-tile body
// is jpeg_tile.h (colour conversion, DCT, quantiser, staging, stores); only the way pixels reach phase A differs, so the
// tuple must be bit-identical — checked before timing.  4:2:0, rows dword aligned, width a multiple of 512, height of 16
// (the experiment's shapes; edges are the product kernel's business).
//
//   build:  make -C pixo_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize \
//             -I pixo_amd/csrc -I include tools/experiments/jpeg_coeffs_persistent.hip pixo_amd/csrc/jpeg_host.cpp \
//             -L pixo_amd -lpixo_hip -Wl,-rpath,'$ORIGIN/../../../pixo_amd' -o tools/ubench/bin/jpeg_coeffs_persistent
//
// Per tile and workgroup (3 wavefronts, as in the product):
//   wait for the tile's 24 KiB of pixels in the raw slot (a counted vmcnt: the wavefront's own 8 DMA instructions)
//   | barrier | phase A: 12 + 12 bytes per lane and item from the raw slot (LDS) -> colour conversion -> planar LDS
//   | barrier | issue the NEXT tile's 8 DMA instructions per wavefront into the raw slot (free now)
//   | phase B exactly as the product: rows, columns, quantiser, stage, 1 KiB stores | barrier (the planar area is free)
// LDS per workgroup: 24 KiB raw slot + 16.5 KiB planar = 41 KiB: three workgroups (nine wavefronts) per CU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "jpeg_host.hpp"
#include "jpeg_tile.h"
#include "pixo_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using namespace pixo_tile;

__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// one wave-instruction: 64 lanes x 16 bytes from global memory to LDS at m0 + lane * 16 (non-temporal: read once)
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

constexpr int kRawBytes = 16 * 1536; // a tile's pixels: 16 rows x 512 px x 3 bytes

struct PArgs {
    const uint8_t *px; uint32_t W, H;
    int16_t *y, *cb, *cr;
    const float *qt;
    uint32_t tiles_x, tiles, units_x, units_y;
};

template <int SLOTS>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 3))) void persistent_kernel(const PArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[]; // [SLOTS raw slots][planar]
    uint8_t *planar = lds + SLOTS * kRawBytes;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t stride = a.W * 3u;
    TileCtx c;
    c.px = a.px; c.y = a.y; c.cb = a.cb; c.cr = a.cr; c.qt = a.qt; c.W = a.W; c.H = a.H;
    c.units_x = a.units_x; c.units_y = a.units_y; c.fast = 1; c.px_end = a.px + (size_t)stride * a.H;
    const uint32_t raw_lds = (uint32_t)(uintptr_t)lds;
    auto issue = [&](uint32_t t, uint32_t slot) { // this wavefront's third of tile t -> raw slot
        const uint32_t tx = t % a.tiles_x, ty = t / a.tiles_x;
        const uint8_t *tile = a.px + (size_t)ty * 16 * stride + tx * 1536u;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int ch = (wave * 8 + k) * 64 + lane, row = ch / 96, col = (ch % 96) * 16; // 16-byte chunk of the tile, row major
            glds16(tile + (size_t)row * stride + col, raw_lds + slot * kRawBytes + (wave * 8 + k) * 1024);
        }
    };
    uint32_t t = blockIdx.x, it = 0;
    if (t < a.tiles) issue(t, 0); // prologue: the first tile
    for (; t < a.tiles; t += gridDim.x, it++) {
        const uint32_t slot = it % SLOTS;
        // vmcnt retires in issue order.  Behind this tile's 8 DMA instructions (per wavefront) were issued: the 8 stores of the
        // previous tile's phase B (it > 0) and — two slots — the 8 DMAs of the tile ahead, issued here.
        bool ahead_out = false;
        if (SLOTS == 2 && t + gridDim.x < a.tiles) { issue(t + gridDim.x, (it + 1) % 2); ahead_out = true; } // (that slot's tile was converted an iteration ago)
        if (it == 0) { if (ahead_out) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else if (ahead_out) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        lds_barrier(); // every wavefront's part of the tile has landed
        const uint32_t tx = t % a.tiles_x, ty = t / a.tiles_x;
        // ---- phase A from the raw slot
        __builtin_amdgcn_s_setprio(1);
        {
            const uint8_t *raw = lds + slot * kRawBytes;
            constexpr int base = 16 / kWaves, extra = 16 % kWaves; // items 6 / 5 / 5
            const int first = wave < extra ? wave * (base + 1) : extra * (base + 1) + (wave - extra) * base;
            const int count = wave < extra ? base + 1 : base;
            for (int j = 0; j < count; j++) {
                const int k = first + j, g = (k & 1) * 64 + lane, row = k >> 1;
                uint32_t r[6];
                const uint32_t *p0 = reinterpret_cast<const uint32_t *>(raw + (2 * row) * 1536 + 12 * g);
                const uint32_t *p1 = reinterpret_cast<const uint32_t *>(raw + (2 * row + 1) * 1536 + 12 * g);
                r[0] = p0[0]; r[1] = p0[1]; r[2] = p0[2]; r[3] = p1[0]; r[4] = p1[1]; r[5] = p1[2];
                producer_color_item<M420, true>(k, lane, r, planar);
            }
        }
        lds_barrier(); // planar complete; the raw slot is free
        __builtin_amdgcn_s_setprio(0);
        if (SLOTS == 1 && t + gridDim.x < a.tiles) issue(t + gridDim.x, 0);
        // ---- phase B, the product's
        float v[64];
        consumer_rows<M420>(wave, lane, planar, v);
        consumer_cols<true>(v);
        uint8_t *stage = planar + stage_offset<M420>(wave);
        uint32_t qw[32];
        consumer_quant<M420>(wave, lane, a.qt, v, qw);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            consumer_stage_blocks(lane, h, qw, stage);
            consumer_stage_sync();
            consumer_store_blocks<M420>(c, tx, ty, wave, lane, h, stage);
            consumer_stage_sync();
        }
        lds_barrier(); // the planar area (and the stages inside it) may be overwritten
    }
}

typedef int (*coeffs_device_fn)(const void *, uint32_t, uint32_t, uint8_t, uint8_t, uint8_t, uint32_t, void *, void *, void *, void *);

int main(int argc, char **argv)
{
    const uint32_t W = 4096, H = 4096, Q = 80;
    const size_t px_bytes = (size_t)W * H * 3, yb = (size_t)W * H / 64, cbn = yb / 4;
    std::vector<uint8_t> host(px_bytes);
    uint32_t state = 42;
    for (size_t i = 0; i < px_bytes; i++) { state = state * 1103515245u + 12345u; host[i] = (uint8_t)(state >> 16); }
    const int NB = 7;
    uint8_t *in[NB]; int16_t *out[NB], *ref;
    for (int i = 0; i < NB; i++) {
        CK(hipMalloc(&in[i], px_bytes)); CK(hipMalloc(&out[i], (yb + 2 * cbn) * 128));
        for (size_t k = 0; k < px_bytes && i; k += 4099) host[k] ^= (uint8_t)i; // (distinct content per buffer)
        CK(hipMemcpy(in[i], host.data(), px_bytes, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&ref, (yb + 2 * cbn) * 128));
    float qt_host[pixo_host::kDeviceQtFloats];
    pixo_host::fill_device_qt((uint8_t)Q, qt_host);
    float *qt; CK(hipMalloc(&qt, sizeof qt_host)); CK(hipMemcpy(qt, qt_host, sizeof qt_host, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto product = [&](int k, int16_t *o) {
        return pixo_hip_jpeg_coeffs_device(in[k], W, H, 2, 1, (uint8_t)Q, 1, o, o + yb * 64, o + (yb + cbn) * 64, nullptr);
    };
    auto time = [&](const char *name, auto launch) {
        int n = 0;
        for (int i = 0; i < 3000; i++, n++) launch(n % NB);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
        float sum = 0, best = 1e9f;
        const int K = 1000, R = 3;
        for (int r = 0; r < R; r++) {
            (void)hipEventRecord(e0);
            for (int i = 0; i < K; i++, n++) launch(n % NB);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            sum += ms; best = ms < best ? ms : best;
        }
        printf("%-96s %7.2f us per launch (best block %6.2f)  %.3f of 8 TB/s\n", name, sum / R * 1e3 / K, best * 1e3 / K, 100.663296 / (sum / R * 1e3 / K) / 8.0);
        fflush(stdout);
    };
    if (product(0, ref) != 0) { printf("product kernel failed: %s\n", pixo_hip_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<int16_t> want((yb + 2 * cbn) * 64), got((yb + 2 * cbn) * 64);
    CK(hipMemcpy(want.data(), ref, want.size() * 2, hipMemcpyDeviceToHost));
    time("product: one tile per workgroup, 2048 x 192, register loads (pixo_hip_jpeg_coeffs_device)", [&](int k) { (void)product(k, out[k]); });

    PArgs a;
    a.W = W; a.H = H; a.qt = qt; a.tiles_x = W / 512; a.tiles = (W / 512) * (H / 16); a.units_x = W / 16; a.units_y = H / 16;
    auto persistent = [&](auto kernel, int slots, unsigned groups, const char *label) {
        const size_t lds = (size_t)slots * kRawBytes + Geo<M420>::planar;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { printf("%s: LDS size refused\n", label); return; }
        auto launch = [&](int k) {
            PArgs b = a;
            b.px = in[k]; b.y = out[k]; b.cb = out[k] + yb * 64; b.cr = out[k] + (yb + cbn) * 64;
            hipLaunchKernelGGL(kernel, dim3(groups), dim3(kThreads), lds, 0, b);
        };
        (void)hipMemset(out[0], 0xFF, want.size() * 2);
        launch(0);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: failed\n", label); return; }
        (void)hipMemcpy(got.data(), out[0], got.size() * 2, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < got.size(); i++) bad += got[i] != want[i];
        char name[200];
        snprintf(name, sizeof name, "persistent, LDS-DMA, %s: %u workgroups (%u per CU), %d raw slot(s), %zu KiB LDS%s", label, groups, groups / 256, slots, lds >> 10,
                 bad ? "  !! TUPLE DIFFERS" : "  (tuple bit-identical)");
        if (bad) { printf("%s: %zu coefficients differ\n", name, bad); return; }
        time(name, launch);
    };
    persistent(persistent_kernel<1>, 1, 768, "one slot, next tile issued after phase A");
    persistent(persistent_kernel<1>, 1, 512, "one slot");
    persistent(persistent_kernel<1>, 1, 1024, "one slot (4 per CU do not fit: 3 resident)");
    persistent(persistent_kernel<2>, 2, 512, "two slots, one tile ahead");
    persistent(persistent_kernel<2>, 2, 256, "two slots");
    persistent(persistent_kernel<1>, 1, 2048, "one slot, one tile per workgroup (the DMA form of the product's shape, 3 per CU)");
    time("product (again)", [&](int k) { (void)product(k, out[k]); });
    return 0;
}
