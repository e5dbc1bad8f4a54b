// dct_rate.hip — in-situ VALU cost of the real column pass (aan8 on 64 live registers), and of
// v_add_f32 with three DISTINCT registers (valu_rates.hip only used dst == src0).
// Reports ns per wave-instruction per SIMD and the shader clock measured with s_memtime against
// the 100 MHz wall clock, so cycles = ns * GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../pixo_amd/csrc/jpeg_tile.h"
#pragma clang fp contract(off)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace pixo_tile;
constexpr int ITERS = 400;

template <int UNROLL> __global__ __launch_bounds__(256) void cols(float *out, float seed, unsigned long long *clk)
{
    float v[64];
    for (int i = 0; i < 64; i++) v[i] = seed + threadIdx.x * 0.001f + i;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < ITERS / UNROLL; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            block_cols(v);
            for (int i = 0; i < 64; i++) v[i] *= 0.125f; // keeps values finite (64 more full-rate ops)
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 64; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

// 8 independent adds, three distinct registers each: d_i = a_i + b_i, then rotate
template <int VARIANT> __global__ __launch_bounds__(256) void add3(float *out, float seed)
{
    float a[8], b[8], d[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x + i; b[i] = seed * i; d[i] = 0; }
    for (int it = 0; it < ITERS * 10; it++) {
        if (VARIANT == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %1, %2" : "=v"(d[i]) : "v"(a[i]), "v"(b[i]));
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(d[i]), "v"(b[i]));
        } else if (VARIANT == 1) { // dst == src0
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        } else if (VARIANT == 2) { // second source a literal constant
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, 0x3f3504f3, %1" : "=v"(d[i]) : "v"(a[i]));
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, 0x3f3504f3, %1" : "=v"(a[i]) : "v"(d[i]));
        } else { // sgpr source
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, %2, %1" : "=v"(d[i]) : "v"(a[i]), "s"(seed));
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, %2, %1" : "=v"(a[i]) : "v"(d[i]), "s"(seed));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    unsigned long long *clk; CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("column pass: 8 x aan8 (29 add/sub + 13 mul) + 64 mul = 400 VALU per iteration\n");
    void (*cfn[4])(float *, float, unsigned long long *) = {cols<1>, cols<8>, cols<16>, cols<40>};
    const int unr[4] = {1, 8, 16, 40};
    for (int k = 0; k < 4; k++)
    for (int w = 1; w <= 4; w++) {
        dim3 grid(cus * w);
        if (w == 1) printf(" loop body = %d x 400 instructions\n", unr[k]);
        hipLaunchKernelGGL(cfn[k], grid, dim3(256), 0, 0, out, 1.0f, clk);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(cfn[k], grid, dim3(256), 0, 0, out, 1.0f + r, clk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        const double ghz = (double)h[0] / ((double)h[1] * 10.0); // wall clock = 100 MHz
        const double ns = ms * 1e6 / ((double)ITERS * 400 * w);
        printf("  waves/SIMD %d: %.3f ns per wave-instr per SIMD, clock %.2f GHz -> %.2f cycles\n", w, ns, ghz, ns * ghz);
    }
    void (*fns[4])(float *, float) = {add3<0>, add3<1>, add3<2>, add3<3>};
    const char *names[4] = {"v_add_f32 d, a, b (3 distinct)", "v_add_f32 a, a, b (dst = src0)", "v_mul_f32 d, literal, a", "v_mul_f32 d, sgpr, a"};
    for (int k = 0; k < 4; k++) {
        for (int w = 2; w <= 4; w += 2) {
            dim3 grid(cus * w);
            hipLaunchKernelGGL(fns[k], grid, dim3(256), 0, 0, out, 1.0f);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(fns[k], grid, dim3(256), 0, 0, out, 1.0f + r);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
            printf("  %-34s waves/SIMD %d: %.3f ns per wave-instr per SIMD\n", names[k], w, ms * 1e6 / ((double)ITERS * 10 * 16 * w));
        }
    }
    return 0;
}
