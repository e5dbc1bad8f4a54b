// valu_ilp.hip — how much ILP x TLP saturates the gfx950 VALU: cycles per wave64 v_add_f32 for
// K independent dependency chains per wave (K = 1, 2, 4, 8) at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4000;

template <int K> __global__ __launch_bounds__(256) void chain(float *out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = seed;
    for (int i = 0; i < ITERS; i++) {
        if (K == 1) asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (K == 2) asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (K == 4) asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (K == 8) asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("cycles (at 2.4 GHz nominal) per wave64 v_add_f32 PER SIMD;  rows: chains per wave, cols: waves per SIMD\n      ");
    for (int w = 1; w <= 6; w++) printf("  w=%d   ", w);
    printf("\n");
    void (*fns[4])(float *, float) = {chain<1>, chain<2>, chain<4>, chain<8>};
    const int ks[4] = {1, 2, 4, 8};
    for (int k = 0; k < 4; k++) {
        printf("K=%d   ", ks[k]);
        for (int w = 1; w <= 6; w++) {
            dim3 grid(cus * w);
            hipLaunchKernelGGL(fns[k], grid, dim3(256), 0, 0, out, 1.0f);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(fns[k], grid, dim3(256), 0, 0, out, 1.0f + r);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
            double inst_per_simd = (double)ITERS * 8 * w;
            printf("%7.2f ", ms * 1e6 / inst_per_simd * 2.4);
        }
        printf("\n");
    }
    return 0;
}
