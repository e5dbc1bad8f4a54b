// ifetch.hip — does straight-line code size limit VALU issue on gfx950?  A loop whose body is N
// independent-ish v_add_f32 (8 B VOP3 encoding forced with a literal... here VOP2 4 B + VOP3 8 B mix)
// is run at 2 waves/SIMD; per-instruction time vs body size shows the instruction-cache knee.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define I8 "v_add_f32 %0, %0, %8\n v_mul_f32 %1, 0x3f800001, %1\n v_add_f32 %2, %2, %8\n v_mul_f32 %3, 0x3f800001, %3\n v_add_f32 %4, %4, %8\n v_mul_f32 %5, 0x3f800001, %5\n v_add_f32 %6, %6, %8\n v_mul_f32 %7, 0x3f800001, %7\n"
#define I64 I8 I8 I8 I8 I8 I8 I8 I8
#define I512 I64 I64 I64 I64 I64 I64 I64 I64

template <int REP> __global__ __launch_bounds__(256) void body(float *out, float seed, int iters)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = seed;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < REP; r++)
            asm volatile(I512 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int REP> int run(float *out, int cus, hipEvent_t e0, hipEvent_t e1)
{
    const int total_inst = 1 << 21; // per wave
    const int iters = total_inst / (512 * REP);
    for (int w : {1, 2, 4}) {
        dim3 grid(cus * w);
        hipLaunchKernelGGL(body<REP>, grid, dim3(256), 0, 0, out, 1.0f, iters);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(body<REP>, grid, dim3(256), 0, 0, out, 2.0f, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("body %6d instr (~%4d KB)  waves/SIMD %d : %6.2f cyc@2.4GHz per instr per SIMD\n", 512 * REP, 512 * REP * 6 / 1024, w,
               ms * 1e6 / ((double)iters * 512 * REP * w) * 2.4);
    }
    return 0;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    run<1>(out, cus, e0, e1); run<2>(out, cus, e0, e1); run<4>(out, cus, e0, e1); run<8>(out, cus, e0, e1);
    run<16>(out, cus, e0, e1); run<32>(out, cus, e0, e1);
    return 0;
}
