// launch_overhead.hip — what a kernel launch costs by itself in the bench's timing method
// (K back-to-back launches on one stream between two events): empty kernels of several grid
// sizes, and a kernel that only writes 50 MB (end-of-kernel write-back included).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void empty(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ __launch_bounds__(192) void lds_only(int *p)
{
    __shared__ int s[4224];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (s[(threadIdx.x + 1) % 192] == 9999) *p = 1;
}
__global__ __launch_bounds__(192) void write_only(uint4 *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * 192 + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * 192) out[i] = make_uint4(i, 1, 2, 3);
}
int main()
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint4 *buf; const size_t n = 50331648 / 16; CK(hipMalloc(&buf, n * 16));
    const int K = 200;
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 20; i++) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < K; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.2f us per launch\n", name, ms * 1e3 / K);
    };
    time("empty kernel, 1 x 64", [&] { hipLaunchKernelGGL(empty, dim3(1), dim3(64), 0, 0, nullptr); });
    time("empty kernel, 2048 x 192", [&] { hipLaunchKernelGGL(empty, dim3(2048), dim3(192), 0, 0, nullptr); });
    time("empty kernel, 16384 x 192", [&] { hipLaunchKernelGGL(empty, dim3(16384), dim3(192), 0, 0, nullptr); });
    time("17 KB LDS + barrier, 2048 x 192", [&] { hipLaunchKernelGGL(lds_only, dim3(2048), dim3(192), 0, 0, nullptr); });
    time("write 50 MB, 2048 x 192", [&] { hipLaunchKernelGGL(write_only, dim3(2048), dim3(192), 0, 0, buf, n); });
    return 0;
}
