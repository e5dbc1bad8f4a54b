// dispatch_gate.cpp — see dispatch_gate.hpp.
#include "dispatch_gate.hpp"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>

namespace pixo_dev {
namespace {
constexpr int kRing = 16, kSlots = 8, kMaxDevices = 64;
struct DeviceGate {
    std::mutex m;
    unsigned long long *ring = nullptr; // pinned: kRing x kSlots words
    bool ring_failed = false;
    unsigned long long seq = 0;
    hipStream_t stream_of[kRing] = {};          // launch n's stream and workgroups at n % kRing
    unsigned long long workgroups_of[kRing] = {};
    unsigned long long room = 0;                // workgroups of single-pass kernels the device surely holds at once
};
DeviceGate g_gates[kMaxDevices];
std::atomic<unsigned long long> g_waits{0}, g_timeouts{0};
} // namespace

DispatchGate::DispatchGate(hipStream_t s, unsigned long long workgroups)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    DeviceGate &g = g_gates[dev % kMaxDevices];
    g.m.lock();
    gate_ = &g;
    if (!g.ring && !g.ring_failed) {
        void *p = nullptr;
        if (hipHostMalloc(&p, sizeof(unsigned long long) * kRing * kSlots, hipHostMallocDefault) == hipSuccess) {
            std::memset(p, 0, sizeof(unsigned long long) * kRing * kSlots);
            g.ring = static_cast<unsigned long long *>(p);
        } else {
            (void)hipGetLastError();
            g.ring_failed = true; // (no gate: the bounded waits and the multi-pass kernels remain)
        }
    }
    if (!g.ring) return;
    if (!g.room) { // (every single-pass kernel fits at least four workgroups on a CU: 19 KiB of LDS, 3-4 wavefronts)
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 1; }
        g.room = 4ull * static_cast<unsigned long long>(cus);
    }
    // Launches of OTHER streams that are not known to be fully dispatched yet (the last kRing - 1 launches are looked at; what is older
    // has long started): while their workgroups and this launch's do not all fit the device together, this launch waits.  A launch
    // whose workgroups fit beside them cannot starve anybody — everything that waits for something can be resident at once.
    auto pending = [&] {
        unsigned long long sum = 0;
        for (unsigned long long n = g.seq; n != 0 && n + kRing > g.seq + 1; --n) {
            if (g.stream_of[n % kRing] == s) continue; // (ordered in front of this launch by the stream)
            const volatile unsigned long long *p = g.ring + (n % kRing) * kSlots;
            bool there = true;
            for (int j = 0; j < kSlots; j++) there = there && p[j] == n;
            if (!there) sum += g.workgroups_of[n % kRing];
        }
        return sum;
    };
    if (pending() != 0 && pending() + workgroups > g.room) {
        g_waits.fetch_add(1, std::memory_order_relaxed);
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(5);
        unsigned spins = 0;
        for (unsigned long long left = pending(); left != 0 && left + workgroups > g.room; left = pending()) {
            if ((++spins & 63u) == 0) {
                if (std::chrono::steady_clock::now() > deadline) { g_timeouts.fetch_add(1, std::memory_order_relaxed); break; }
                std::this_thread::yield();
            }
        }
    }
    ++g.seq;
    g.stream_of[g.seq % kRing] = s;
    g.workgroups_of[g.seq % kRing] = workgroups;
    m_.slots = g.ring + (g.seq % kRing) * kSlots;
    m_.seq = g.seq;
}

DispatchGate::~DispatchGate()
{
    if (gate_) static_cast<DeviceGate *>(gate_)->m.unlock();
}

void dispatch_gate_stats(unsigned long long *waits, unsigned long long *timeouts)
{
    if (waits) *waits = g_waits.load(std::memory_order_relaxed);
    if (timeouts) *timeouts = g_timeouts.load(std::memory_order_relaxed);
}

} // namespace pixo_dev
