// phase_rate.hip — saturated (VALU-bound) cost of each phase of the shipped tile body, measured on
// the real device functions of jpeg_tile.h with no HBM traffic: W workgroups per CU run the phase
// in a loop on LDS-resident data.  Reports shader cycles per (wave, tile) and the equivalent
// whole-image time for a 4096x4096 4:2:0 image (2048 tiles over 256 CUs x 4 SIMDs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../pixo_amd/csrc/jpeg_tile.h"
#pragma clang fp contract(off)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace pixo_tile;
constexpr int ITERS = 64;
enum { P_COLOR = 1, P_ROWS = 2, P_COLS = 4, P_QUANT = 8, P_READBACK = 16 };

template <int PH> __global__ __launch_bounds__(256) void run(const float *qt, uint32_t *out, uint32_t seed, unsigned long long *clk)
{
    typedef Geo<M420> G;
    __shared__ __attribute__((aligned(16))) uint8_t lds[G::planar];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < G::planar / 4; i += 256) ((uint32_t *)lds)[i] = (i * 2654435761u + seed) & 0x7f7f7f7fu;
    __syncthreads();
    constexpr int Q = G::items / 4;
    uint32_t r[Q * G::item_regs];
    for (int i = 0; i < Q * G::item_regs; i++) r[i] = lane * 77 + i + seed;
    float v[64];
    for (int i = 0; i < 64; i++) v[i] = (float)(lane + i);
    uint32_t acc = 0;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    const int bw = wave < 3 ? wave : 1; // all four waves do consumer work here (wave 3 repeats wave 1's block kind)
    uint8_t *stage = lds + stage_offset<M420>(bw);
    for (int it = 0; it < ITERS; it++) {
        if (PH & P_COLOR) {
#pragma unroll
            for (int j = 0; j < Q; j++) {
                for (int i = 0; i < G::item_regs; i++) asm volatile("" : "+v"(r[j * G::item_regs + i]));
                producer_color_item<M420>(wave * Q + j, lane, &r[j * G::item_regs], lds);
            }
        }
        if (PH & P_ROWS) consumer_rows<M420>(bw, lane, lds, v);
        if (PH & P_COLS) consumer_cols(v);
        if (PH & P_QUANT) {
            consumer_quant_half<M420>(bw, lane, qt, v, 0, stage);
            if (PH & P_READBACK) for (int k = 0; k < 4; k++) { const int ch = k * 64 + lane; const u32x4 w = *(const u32x4 *)(stage + stage_addr(ch >> 2, ch & 3)); acc ^= w.x ^ w.y ^ w.z ^ w.w; }
            consumer_quant_half<M420>(bw, lane, qt, v, 1, stage);
            if (PH & P_READBACK) for (int k = 0; k < 4; k++) { const int ch = k * 64 + lane; const u32x4 w = *(const u32x4 *)(stage + stage_addr(ch >> 2, ch & 3)); acc ^= w.x ^ w.y ^ w.z ^ w.w; }
        }
        if (!(PH & P_ROWS)) for (int i = 0; i < 64; i++) asm volatile("" : "+v"(v[i]));
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 64; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc ^ __builtin_bit_cast(uint32_t, s) ^ lds[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

__global__ __launch_bounds__(256) void stamped(const float *qt, uint32_t *out, uint32_t seed, unsigned long long *st)
{
    typedef Geo<M420> G;
    __shared__ __attribute__((aligned(16))) uint8_t lds[G::planar];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < G::planar / 4; i += 256) ((uint32_t *)lds)[i] = (i * 2654435761u + seed) & 0x7f7f7f7fu;
    __syncthreads();
    float v[64];
    uint32_t acc = 0;
    const int bw = wave < 3 ? wave : 1;
    uint8_t *stage = lds + stage_offset<M420>(bw);
    unsigned long long sum[6] = {0, 0, 0, 0, 0, 0};
    for (int it = 0; it < ITERS; it++) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        consumer_rows<M420>(bw, lane, lds, v);
        for (int i = 0; i < 64; i++) asm volatile("" : "+v"(v[i]));
        const unsigned long long t1 = __builtin_readcyclecounter();
        consumer_cols(v);
        const unsigned long long t2 = __builtin_readcyclecounter();
        consumer_quant_half<M420>(bw, lane, qt, v, 0, stage);
        const unsigned long long t3 = __builtin_readcyclecounter();
        for (int k = 0; k < 4; k++) { const int ch = k * 64 + lane; const u32x4 w = *(const u32x4 *)(stage + stage_addr(ch >> 2, ch & 3)); acc ^= w.x ^ w.y ^ w.z ^ w.w; }
        asm volatile("" : "+v"(acc));
        const unsigned long long t4 = __builtin_readcyclecounter();
        consumer_quant_half<M420>(bw, lane, qt, v, 1, stage);
        const unsigned long long t5 = __builtin_readcyclecounter();
        for (int k = 0; k < 4; k++) { const int ch = k * 64 + lane; const u32x4 w = *(const u32x4 *)(stage + stage_addr(ch >> 2, ch & 3)); acc ^= w.x ^ w.y ^ w.z ^ w.w; }
        asm volatile("" : "+v"(acc));
        const unsigned long long t6 = __builtin_readcyclecounter();
        sum[0] += t1 - t0; sum[1] += t2 - t1; sum[2] += t3 - t2; sum[3] += t4 - t3; sum[4] += t5 - t4; sum[5] += t6 - t5;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (lane == 0 && blockIdx.x == 0) for (int i = 0; i < 6; i++) st[wave * 6 + i] = sum[i] / ITERS;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    float *qt; CK(hipMalloc(&qt, kQtFloats * 4));
    float hq[kQtFloats];
    for (int i = 0; i < kQtFloats; i++) hq[i] = (i < 128 || i >= 256) ? 1.0f / (float)(3 + i % 61) : (float)(3 + i % 61);
    CK(hipMemcpy(qt, hq, sizeof hq, hipMemcpyHostToDevice));
    unsigned long long *clk; CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char *name; void (*fn)(const float *, uint32_t *, uint32_t, unsigned long long *); double waves_per_tile; } ph[] = {
        {"colour (4 items/wave)", run<P_COLOR>, 4},
        {"rows (LDS->reg, row pass)", run<P_ROWS>, 3},
        {"cols", run<P_COLS>, 3},
        {"rows+cols", run<P_ROWS | P_COLS>, 3},
        {"quant (2 halves -> stage)", run<P_QUANT>, 3},
        {"quant + stage read-back", run<P_QUANT | P_READBACK>, 3},
        {"rows+cols+quant+readback", run<P_ROWS | P_COLS | P_QUANT | P_READBACK>, 3},
        {"colour+rows+cols+quant+rb", run<P_COLOR | P_ROWS | P_COLS | P_QUANT | P_READBACK>, 3},
    };
    printf("%-30s %s\n", "phase", "W=wgs/CU: ns per wave-pass per SIMD | image-equivalent us (2048 tiles, waves/tile, 1024 SIMDs)");
    for (auto &p : ph) {
        printf("%-30s", p.name);
        for (int w = 1; w <= 4; w++) {
            dim3 grid(cus * w);
            hipLaunchKernelGGL(p.fn, grid, dim3(256), 0, 0, qt, out, 1u, clk);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(p.fn, grid, dim3(256), 0, 0, qt, out, 2u + r, clk);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
            unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double ghz = (double)h[0] / ((double)h[1] * 10.0);
            const double ns = ms * 1e6 / ((double)ITERS * w); // per wave-pass per SIMD (each SIMD holds w waves)
            printf("  W%d %7.0f ns (%5.0f cyc) %5.2f us", w, ns, ns * ghz, ns * 2048.0 * p.waves_per_tile / 1024.0 / 1e3);
        }
        printf("\n");
    }
    unsigned long long *st; CK(hipMalloc(&st, 24 * 8));
    for (int w = 1; w <= 4; w++) {
        hipLaunchKernelGGL(stamped, dim3(cus * w), dim3(256), 0, 0, qt, out, 1u, st);
        CK(hipDeviceSynchronize());
        unsigned long long h[24]; CK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
        for (int wv = 0; wv < 3; wv += 2)
            printf("stamped W%d wave %d (%s): rows %llu  cols %llu  quant0 %llu  readback0 %llu  quant1 %llu  readback1 %llu cycles\n", w, wv, wv == 2 ? "chroma u16" : "luma",
                   h[wv * 6], h[wv * 6 + 1], h[wv * 6 + 2], h[wv * 6 + 3], h[wv * 6 + 4], h[wv * 6 + 5]);
    }
    return 0;
}
