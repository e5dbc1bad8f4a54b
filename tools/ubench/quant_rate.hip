// quant_rate.hip — cost of quantiser formulations (cycles per 64-coefficient block per wave at
// W workgroups/CU, VALU-bound, LDS stage write + read-back included).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../pixo_amd/csrc/jpeg_tile.h"
#pragma clang fp contract(off)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace pixo_tile;
constexpr int ITERS = 64;
struct alignas(16) f32x4 { float x, y, z, w; };

__device__ __forceinline__ uint32_t readback(const uint8_t *stage, int lane)
{
    uint32_t acc = 0;
    for (int k = 0; k < 4; k++) { const int ch = k * 64 + lane; const u32x4 w = *(const u32x4 *)(stage + stage_addr(ch >> 2, ch & 3)); acc ^= w.x ^ w.y ^ w.z ^ w.w; }
    return acc;
}

// exact path for one row (the reference operation)
__device__ __forceinline__ void exact_row(const float *x, qtab_t q, float scale, float *s)
{
#pragma unroll
    for (int c = 0; c < 8; c++) { float n = __builtin_roundf((x[c] * scale) / q[c]); s[c] = n + kRoundMagic; PIXO_SCHED_FENCE(); }
}

template <int VAR> __device__ __forceinline__ void quant_block(const float *v, const float *qt, const float *lds_rcp, int lane, uint8_t *stage, uint32_t &sink)
{
    const qtab_t tab = as_qtab(qt);
    if (VAR == 0) { // shipped
        consumer_quant_half<M420>(0, lane, qt, v, 0, stage); sink ^= readback(stage, lane);
        consumer_quant_half<M420>(0, lane, qt, v, 1, stage); sink ^= readback(stage, lane);
        return;
    }
#pragma unroll
    for (int half = 0; half < 2; half++) {
        float s[32];
        uint32_t acc = 0x80000000u;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int u = half * 4 + r;
            float rc[8];
            if (VAR == 1 || VAR == 2) { // SGPR reciprocals
#pragma unroll
                for (int c = 0; c < 8; c++) rc[c] = tab[u * 8 + c];
            } else { // VGPR reciprocals via LDS broadcast reads
                const f32x4 a = *(const f32x4 *)(lds_rcp + u * 8), b = *(const f32x4 *)(lds_rcp + u * 8 + 4);
                rc[0] = a.x; rc[1] = a.y; rc[2] = a.z; rc[3] = a.w; rc[4] = b.x; rc[5] = b.y; rc[6] = b.z; rc[7] = b.w;
            }
            uint32_t racc = 0x80000000u;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                float rr = v[u * 8 + c] * rc[c];
                s[r * 8 + c] = rr + kRoundMagic;
                if (VAR != 1) {
                    float n = s[r * 8 + c] - kRoundMagic;
                    float d = rr - n;
                    float w = __builtin_fmaf(__builtin_fabsf(rr), 0x1p-21f, __builtin_fabsf(d)) - 0.5f;
                    racc &= fbits(w);
                }
            }
            if (VAR == 2 || VAR == 3) { // branch per row
                if ((racc & 0x80000000u) == 0) exact_row(&v[u * 8], tab + 128 + u * 8, 1.0f, &s[r * 8]);
            } else acc &= racc;
        }
        if (VAR == 4) { // branch per half
            if ((acc & 0x80000000u) == 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) exact_row(&v[(half * 4 + r) * 8], tab + 128 + (half * 4 + r) * 8, 1.0f, &s[r * 8]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            u32x4 o;
            o.x = perm(fbits(s[r * 8 + 1]), fbits(s[r * 8 + 0]), 0x05040100u); o.y = perm(fbits(s[r * 8 + 3]), fbits(s[r * 8 + 2]), 0x05040100u);
            o.z = perm(fbits(s[r * 8 + 5]), fbits(s[r * 8 + 4]), 0x05040100u); o.w = perm(fbits(s[r * 8 + 7]), fbits(s[r * 8 + 6]), 0x05040100u);
            *(u32x4 *)(stage + stage_addr(lane, r)) = o;
        }
        sink ^= readback(stage, lane);
    }
}

template <int VAR> __global__ __launch_bounds__(256) void run(const float *qt, uint32_t *out, uint32_t seed, unsigned long long *clk)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 4096 + 1024];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float *lds_rcp = (float *)(lds + 4 * 4096);
    if (threadIdx.x < 64) lds_rcp[threadIdx.x] = qt[threadIdx.x];
    __syncthreads();
    float v[64];
    for (int i = 0; i < 64; i++) { uint32_t h = (lane * 64 + i + seed) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; v[i] = ((float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 417.3f; }
    uint32_t sink = 0;
    uint8_t *stage = lds + wave * 4096;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < ITERS; it++) {
        for (int i = 0; i < 64; i++) asm volatile("" : "+v"(v[i]));
        quant_block<VAR>(v, qt, lds_rcp, lane, stage, sink);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = sink;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    float *qt; CK(hipMalloc(&qt, kQtFloats * 4));
    float hq[kQtFloats];
    for (int i = 0; i < kQtFloats; i++) { const float q = (float)(3 + (i * 7) % 61); hq[i] = (i < 128 || i >= 256) ? 1.0f / q : q; }
    for (int i = 0; i < 64; i++) { const float q = (float)(3 + (i * 7) % 61); hq[i] = 1.0f / q; hq[128 + i] = q; }
    CK(hipMemcpy(qt, hq, sizeof hq, hipMemcpyHostToDevice));
    unsigned long long *clk; CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char *name; void (*fn)(const float *, uint32_t *, uint32_t, unsigned long long *); } ph[] = {
        {"0 shipped (SGPR rcp, branch/row)", run<0>},
        {"1 floor: mul + magic + pack only", run<1>},
        {"2 SGPR rcp, branch/row (rewritten)", run<2>},
        {"3 VGPR rcp via LDS, branch/row", run<3>},
        {"4 VGPR rcp via LDS, branch/half", run<4>},
    };
    for (auto &p : ph) {
        printf("%-38s", p.name);
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
            const double ns = ms * 1e6 / ((double)ITERS * w);
            printf("  W%d %6.0f cyc", w, ns * ghz);
        }
        printf("\n");
    }
    return 0;
}
