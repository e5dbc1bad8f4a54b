// duplex.cpp — do a host-to-device copy and a device-to-host copy on two streams overlap (PCIe is full duplex)?  Pageable and pinned host
// memory on either side, copies enqueued from one host thread.   hipcc -O2 tools/ubench/duplex.cpp -o /tmp/duplex && /tmp/duplex
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t IN = 48u << 20, OUT = 11u << 20; // a 4096x4096 RGB image in, its JPEG file out
    void *d_in, *d_out, *pin_in, *pin_out;
    CK(hipMalloc(&d_in, IN)); CK(hipMalloc(&d_out, OUT));
    CK(hipHostMalloc(&pin_in, IN, hipHostMallocDefault)); CK(hipHostMalloc(&pin_out, OUT, hipHostMallocDefault));
    void *pg_in = malloc(IN), *pg_out = malloc(OUT);
    memset(pg_in, 1, IN); memset(pg_out, 1, OUT); memset(pin_in, 1, IN); memset(pin_out, 1, OUT);
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    auto run = [&](const char *name, void *hin, void *hout, int mode) -> int { // mode 0: in only, 1: out only, 2: both (in first), 3: both, 8 bands in / 8 pieces out interleaved, 4: both from two host threads
        double best = 1e9;
        for (int rep = 0; rep < 7; rep++) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            if (mode == 0 || mode == 2) CK(hipMemcpyAsync(d_in, hin, IN, hipMemcpyHostToDevice, a));
            if (mode == 1 || mode == 2) CK(hipMemcpyAsync(hout, d_out, OUT, hipMemcpyDeviceToHost, b));
            if (mode == 3)
                for (int k = 0; k < 8; k++) {
                    CK(hipMemcpyAsync((char *)d_in + k * (IN / 8), (char *)hin + k * (IN / 8), IN / 8, hipMemcpyHostToDevice, a));
                    CK(hipMemcpyAsync((char *)hout + k * (OUT / 8), (char *)d_out + k * (OUT / 8), OUT / 8, hipMemcpyDeviceToHost, b));
                }
            if (mode == 4) {
                std::thread t([&] { (void)hipMemcpyAsync(hout, d_out, OUT, hipMemcpyDeviceToHost, b); (void)hipStreamSynchronize(b); });
                CK(hipMemcpyAsync(d_in, hin, IN, hipMemcpyHostToDevice, a));
                CK(hipStreamSynchronize(a));
                t.join();
            }
            CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
            const double dt = now() - t0;
            if (dt < best) best = dt;
        }
        printf("%-44s %-34s %7.3f ms\n", name, mode == 0 ? "48 MiB in" : mode == 1 ? "11 MiB out" : mode == 2 ? "both, two streams" : mode == 3 ? "both, 8 bands interleaved" : "both, two host threads", best);
        return 0;
    };
    for (int mode = 0; mode < 5; mode++) {
        if (run("pinned in, pinned out", pin_in, pin_out, mode)) return 1;
        if (run("pageable in, pinned out", pg_in, pin_out, mode)) return 1;
        if (run("pageable in, pageable out", pg_in, pg_out, mode)) return 1;
    }
    return 0;
}
