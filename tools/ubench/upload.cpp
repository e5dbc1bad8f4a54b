// Microbenchmark: ways of getting 48 MiB of pageable host pixels into HBM.
//   hipcc -O2 -o /tmp/upload tools/ubench/upload.cpp -lpthread && /tmp/upload
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t n = size_t{48} << 20;
    uint8_t *page = static_cast<uint8_t *>(std::malloc(n)), *pin = nullptr, *dev = nullptr;
    std::memset(page, 7, n);
    CK(hipHostMalloc(reinterpret_cast<void **>(&pin), n, hipHostMallocDefault));
    CK(hipMalloc(reinterpret_cast<void **>(&dev), n));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2] * 1e3; };
    std::vector<double> t;
    for (int i = 0; i < 9; i++) { double a = now(); CK(hipMemcpyAsync(dev, page, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); t.push_back(now() - a); }
    std::printf("pageable hipMemcpyAsync            %.3f ms\n", med(t)); t.clear();
    for (int i = 0; i < 9; i++) { double a = now(); CK(hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); t.push_back(now() - a); }
    std::printf("pinned hipMemcpyAsync              %.3f ms\n", med(t)); t.clear();
    for (int threads : {1, 2, 4, 8, 16}) {
        for (int i = 0; i < 9; i++) {
            double a = now();
            std::vector<std::thread> w;
            for (int k = 0; k < threads; k++) w.emplace_back([&, k] { size_t lo = n * k / threads, hi = n * (k + 1) / threads; std::memcpy(pin + lo, page + lo, hi - lo); });
            for (auto &x : w) x.join();
            t.push_back(now() - a);
        }
        std::printf("memcpy pageable->pinned, %2d threads %.3f ms\n", threads, med(t)); t.clear();
    }
    // pipeline: T threads copy 2 MiB chunks in order; the main thread issues the DMA of each chunk as it becomes ready
    for (int threads : {2, 4, 8}) for (size_t chunk : {size_t{1} << 20, size_t{4} << 20}) {
        const size_t nchunks = (n + chunk - 1) / chunk;
        for (int i = 0; i < 9; i++) {
            double a = now();
            std::vector<std::atomic<int>> ready(nchunks);
            for (auto &r : ready) r.store(0);
            std::atomic<size_t> next{0};
            std::vector<std::thread> w;
            for (int k = 0; k < threads; k++) w.emplace_back([&] {
                for (;;) { size_t c = next.fetch_add(1); if (c >= nchunks) break; size_t lo = c * chunk, hi = std::min(n, lo + chunk);
                           std::memcpy(pin + lo, page + lo, hi - lo); ready[c].store(1, std::memory_order_release); } });
            for (size_t c = 0; c < nchunks; c++) {
                while (!ready[c].load(std::memory_order_acquire)) std::this_thread::yield();
                size_t lo = c * chunk, hi = std::min(n, lo + chunk);
                CK(hipMemcpyAsync(dev + lo, pin + lo, hi - lo, hipMemcpyHostToDevice, s));
            }
            CK(hipStreamSynchronize(s));
            for (auto &x : w) x.join();
            t.push_back(now() - a);
        }
        std::printf("pipeline %d threads, %zu MiB chunks      %.3f ms\n", threads, chunk >> 20, med(t)); t.clear();
    }
    // registering the caller's pages instead
    for (int i = 0; i < 5; i++) { double a = now(); CK(hipHostRegister(page, n, hipHostRegisterDefault)); double b = now(); CK(hipMemcpyAsync(dev, page, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double c2 = now(); CK(hipHostUnregister(page)); double d = now(); if (i == 4) std::printf("register %.3f ms, copy %.3f ms, unregister %.3f ms\n", (b - a) * 1e3, (c2 - b) * 1e3, (d - c2) * 1e3); }
    // ---- the other direction: an 11 MB file from HBM into host memory the caller will own
    const size_t m = size_t{11} << 20;
    for (int i = 0; i < 9; i++) { double a = now(); CK(hipMemcpyAsync(pin, dev, m, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t.push_back(now() - a); }
    std::printf("D2H 11 MiB into pinned                         %.3f ms\n", med(t)); t.clear();
    for (int i = 0; i < 9; i++) { double a = now(); CK(hipMemcpyAsync(page, dev, m, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t.push_back(now() - a); }
    std::printf("D2H 11 MiB into the same pageable buffer       %.3f ms\n", med(t)); t.clear();
    for (int i = 0; i < 9; i++) { double a = now(); uint8_t *f = static_cast<uint8_t *>(std::malloc(m)); CK(hipMemcpyAsync(f, dev, m, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t.push_back(now() - a); std::free(f); }
    std::printf("D2H 11 MiB into malloc, freed every time       %.3f ms (max %.3f)\n", med(t), *std::max_element(t.begin(), t.end()) * 1e3); t.clear();
    { std::vector<uint8_t *> keep;
      for (int i = 0; i < 9; i++) { double a = now(); uint8_t *f = static_cast<uint8_t *>(std::malloc(m)); CK(hipMemcpyAsync(f, dev, m, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t.push_back(now() - a); keep.push_back(f); }
      std::printf("D2H 11 MiB into malloc, all kept (new pages)   %.3f ms (max %.3f)\n", med(t), *std::max_element(t.begin(), t.end()) * 1e3); t.clear();
      for (int i = 0; i < 9; i++) { double a = now(); uint8_t *f = static_cast<uint8_t *>(std::malloc(m)); CK(hipMemcpyAsync(pin, dev, m, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); std::memcpy(f, pin, m); t.push_back(now() - a); keep.push_back(f); }
      std::printf("D2H into pinned + memcpy into kept malloc      %.3f ms (max %.3f)\n", med(t), *std::max_element(t.begin(), t.end()) * 1e3); t.clear();
      for (int i = 0; i < 9; i++) { double a = now(); uint8_t *f = static_cast<uint8_t *>(std::malloc(m)); CK(hipMemcpyAsync(pin, dev, m, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
          std::vector<std::thread> w; for (int k = 0; k < 8; k++) w.emplace_back([&, k] { size_t lo = m * k / 8, hi = m * (k + 1) / 8; std::memcpy(f + lo, pin + lo, hi - lo); }); for (auto &x : w) x.join();
          t.push_back(now() - a); keep.push_back(f); }
      std::printf("D2H into pinned + 8-thread memcpy, kept malloc %.3f ms (max %.3f)\n", med(t), *std::max_element(t.begin(), t.end()) * 1e3); t.clear();
      for (auto f : keep) std::free(f); }
    std::printf("hardware threads: %u\n", std::thread::hardware_concurrency());
    return 0;
}
