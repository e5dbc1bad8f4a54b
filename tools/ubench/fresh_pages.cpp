// fresh_pages.cpp — what does it cost on this box to get N MB of FRESH, resident, caller-owned host memory and to fill it from a warm
// source?  malloc + one-thread memcpy, several threads, MADV_HUGEPAGE (THP in madvise mode), MAP_POPULATE, a recycled block.
//   g++ -O2 -pthread tools/ubench/fresh_pages.cpp -o /tmp/fresh_pages && /tmp/fresh_pages [MB]
#include <sys/mman.h>
#include <algorithm>
#include <chrono>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par(unsigned t, size_t n, const std::function<void(size_t, size_t)> &f)
{
    std::vector<std::thread> th;
    const size_t per = (n + t - 1) / t / (2u << 20) * (2u << 20) + (2u << 20);
    for (unsigned i = 0; i < t; i++) { const size_t a = std::min(n, i * per), b = std::min(n, (i + 1) * per); if (b > a) th.emplace_back([=, &f] { f(a, b); }); }
    for (auto &x : th) x.join();
}
int main(int argc, char **argv)
{
    const size_t n = (size_t)(argc > 1 ? atoi(argv[1]) : 178) << 20;
    char *src = (char *)malloc(n); memset(src, 3, n);
    auto report = [&](const char *name, double alloc, double fill) { printf("%-58s alloc+touch %7.2f ms  fill %7.2f ms  total %7.2f ms\n", name, alloc, fill, alloc + fill); };
    for (int rep = 0; rep < 2; rep++) {
        { double t0 = now(); char *p = (char *)malloc(n); double t1 = now(); memcpy(p, src, n); double t2 = now(); report("malloc, memcpy by 1 thread", t1 - t0, t2 - t1); free(p); }
        for (unsigned t : {4u, 8u, 16u}) { double t0 = now(); char *p = (char *)malloc(n); double t1 = now(); par(t, n, [&](size_t a, size_t b) { memcpy(p + a, src + a, b - a); }); double t2 = now();
            char nm[80]; snprintf(nm, 80, "malloc, memcpy by %u threads", t); report(nm, t1 - t0, t2 - t1); free(p); }
        for (unsigned t : {1u, 8u}) { double t0 = now(); void *q; if (posix_memalign(&q, 2u << 20, (n + (2u << 20) - 1) / (2u << 20) * (2u << 20))) return 1; madvise(q, n, MADV_HUGEPAGE); char *p = (char *)q; double t1 = now();
            par(t, n, [&](size_t a, size_t b) { memcpy(p + a, src + a, b - a); }); double t2 = now(); char nm[80]; snprintf(nm, 80, "2 MiB aligned + MADV_HUGEPAGE, memcpy by %u threads", t); report(nm, t1 - t0, t2 - t1); free(p); }
        { double t0 = now(); char *p = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0); double t1 = now(); memcpy(p, src, n); double t2 = now();
          report("mmap MAP_POPULATE, memcpy by 1 thread", t1 - t0, t2 - t1); munmap(p, n); }
        { double t0 = now(); char *p = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0); double t1 = now(); par(8, n, [&](size_t a, size_t b) { memcpy(p + a, src + a, b - a); }); double t2 = now();
          report("mmap MAP_POPULATE, memcpy by 8 threads", t1 - t0, t2 - t1); munmap(p, n); }
        { char *p = (char *)malloc(n); memset(p, 1, n); double t1 = now(); memcpy(p, src, n); double t2 = now(); report("a RECYCLED block (resident), memcpy by 1 thread", 0, t2 - t1);
          t1 = now(); par(8, n, [&](size_t a, size_t b) { memcpy(p + a, src + a, b - a); }); t2 = now(); report("a RECYCLED block (resident), memcpy by 8 threads", 0, t2 - t1); free(p); }
        puts("");
    }
    return 0;
}
