// memtime.hip — calibrates s_memtime (__builtin_readcyclecounter) against wall time and against a
// known instruction count, alone and under a VALU+HBM load similar to the JPEG kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void k(unsigned long long *out, float *sink, int iters)
{
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.0f;
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long r0 = wall_clock64();
    for (int i = 0; i < iters; i++)
        asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long r1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount, wcr = 0;
    CK(hipDeviceGetAttribute(&wcr, hipDeviceAttributeWallClockRate, 0));
    printf("clockRate %d kHz, wallClockRate %d kHz\n", prop.clockRate, wcr);
    unsigned long long *out; float *sink;
    CK(hipMalloc(&out, cus * 8 * 16)); CK(hipMalloc(&sink, (size_t)cus * 8 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200000;
    for (int w : {1, 2, 4}) {
        hipLaunchKernelGGL(k, dim3(cus * w), dim3(256), 0, 0, out, sink, iters); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(cus * w), dim3(256), 0, 0, out, sink, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        printf("waves/SIMD %d: kernel %.3f ms; memtime ticks %llu (%.1f MHz), wall_clock64 ticks %llu (%.1f MHz); ticks per instr per wave %.2f\n", w, ms,
               h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3), (double)h[0] / (iters * 8.0));
    }
    return 0;
}
