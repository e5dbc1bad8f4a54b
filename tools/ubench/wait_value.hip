// Does hipStreamWaitValue64 work here, on memory a KERNEL writes, and how long after the write does the waiting stream's next kernel start?
// (hipcc --offload-arch=gfx950 -O2 tools/ubench/wait_value.hip -o /tmp/wait_value && timeout 30 /tmp/wait_value)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin_then_add(unsigned long long *sig, unsigned long long *stamp, int spin_us)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        stamp[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        __hip_atomic_fetch_add(sig, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void stamp_now(unsigned long long *stamp) { if (threadIdx.x == 0) stamp[0] = __builtin_amdgcn_s_memrealtime(); }
int main()
{
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned long long *sig = nullptr, *stamps = nullptr, *h = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(hipMallocSignalMemory) -> %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    CK(hipMalloc((void **)&stamps, 64 * 8));
    CK(hipHostMalloc((void **)&h, 64 * 8, hipHostMallocDefault));
    CK(hipMemset(sig, 0, 8));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    for (int rep = 0; rep < 5; rep++) {
        const unsigned long long target = 8ull * (rep + 1);
        CK(hipMemset(stamps, 0, 64 * 8));
        CK(hipDeviceSynchronize());
        // stream b waits for 8 marks, stream a's kernel (8 workgroups) spins 200 us and marks
        CK(hipStreamWaitValue64(b, sig, target, hipStreamWaitValueGte, ~0ull));
        hipLaunchKernelGGL(stamp_now, dim3(1), dim3(64), 0, b, stamps + 32);
        hipLaunchKernelGGL(spin_then_add, dim3(8), dim3(64), 0, a, sig, stamps, 200);
        CK(hipStreamSynchronize(a));
        CK(hipStreamSynchronize(b));
        CK(hipMemcpy(h, stamps, 64 * 8, hipMemcpyDeviceToHost));
        unsigned long long last = 0;
        for (int i = 0; i < 8; i++) last = h[i] > last ? h[i] : last;
        printf("rep %d: waiting stream's kernel started %.2f us after the last mark (negative = the wait did not hold)\n", rep, ((double)h[32] - (double)last) / 100.0);
    }
    // a wait whose condition already holds: cost of the packet
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, b));
    for (int i = 0; i < 100; i++) { CK(hipStreamWaitValue64(b, sig, 1, hipStreamWaitValueGte, ~0ull)); hipLaunchKernelGGL(stamp_now, dim3(1), dim3(64), 0, b, stamps + 33); }
    CK(hipEventRecord(e1, b)); CK(hipStreamSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventRecord(e0, b));
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(stamp_now, dim3(1), dim3(64), 0, b, stamps + 33);
    CK(hipEventRecord(e1, b)); CK(hipStreamSynchronize(b));
    float ms2 = 0; CK(hipEventElapsedTime(&ms2, e0, e1));
    printf("100 x (satisfied wait + tiny kernel): %.2f us each; 100 x tiny kernel alone: %.2f us each\n", ms * 10, ms2 * 10);
    return 0;
}
