// graph_gap.hip — does a hipGraph of K dependent kernel nodes shorten the gap between launches that a
// stream of K launches shows?  Empty 2048x192 kernels and 50 MB writers, stream vs captured graph.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void empty(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ __launch_bounds__(192) void write_only(uint4 *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * 192 + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * 192) __builtin_nontemporal_store(i, (size_t *)(out + i));
}
int main()
{
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint4 *buf; const size_t n = 50331648 / 16; CK(hipMalloc(&buf, n * 16));
    const int K = 200;
    for (int which = 0; which < 2; which++) {
        auto launch = [&] {
            if (which == 0) hipLaunchKernelGGL(empty, dim3(2048), dim3(192), 0, s, nullptr);
            else hipLaunchKernelGGL(write_only, dim3(2048), dim3(192), 0, s, buf, n);
        };
        for (int i = 0; i < 3000; i++) launch();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < K; i++) launch();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s stream: %7.2f us per launch\n", which ? "write 50 MB 2048x192" : "empty 2048x192", ms * 1e3 / K);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < K; i++) launch();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 5; r++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 5; r++) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s graph : %7.2f us per launch\n", which ? "write 50 MB 2048x192" : "empty 2048x192", ms * 1e3 / (5 * K));
    }
    return 0;
}
