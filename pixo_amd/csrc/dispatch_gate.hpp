// dispatch_gate.hpp — ONE single-pass kernel at a time in its DISPATCH phase per device (round 6).
//
// The single-pass kernels (pixels_code, scan_code, prog_code, stuff_fused) let a workgroup wait for workgroups with LOWER ids of its own
// launch.  That is safe for one launch — the hardware starts a grid's workgroups in increasing order, so whatever a resident workgroup
// waits for is resident or done — and for launches behind each other in one stream.  Two launches from two streams (two calling
// threads, each with its own context) that start TOGETHER on an empty device split its workgroup slots between them: every slot then
// holds a workgroup that waits for one that has no slot — neither launch can go on until a bounded wait gives up (0.75 s) and the host
// codes the file with the multi-pass kernels.  Measured: 2 of 6 fresh processes with three threads (tools/mt_first_calls.py).
//
// The gate: every such launch takes the device's next sequence number under a mutex; while launches of OTHER streams are not fully
// dispatched yet and their workgroups + this launch's do not all fit the device at once (4 per CU, conservatively), the host waits
// (bounded: 5 ms).  Small files of several threads therefore never wait: what fits together cannot starve each other.  "Fully dispatched" is told by the kernel itself: consecutive
// workgroup ids go round-robin to the 8 XCDs and every XCD starts its share in order, so when the launch's LAST EIGHT workgroups have
// started, all have; each of them stores the sequence number into its slot of a pinned ring.  A launch whose workgroups are all resident
// (or done) needs no further slot: the next launch can only take what it gives up.  Costs nothing for one calling thread (same stream:
// no wait; eight 8-byte stores a launch) and keeps what several threads gain — the next file's pixels under this file's look-backs.
#pragma once
#include <hip/hip_runtime_api.h>

namespace pixo_dev {

struct GateMark {
    unsigned long long *slots = nullptr; // 8 words of pinned host memory (null: no gate)
    unsigned long long seq = 0;
};

// Constructed right in front of the enqueue of a single-pass kernel on `s`, destroyed right behind it.
class DispatchGate {
  public:
    DispatchGate(hipStream_t s, unsigned long long workgroups); // workgroups: of the launch that follows
    ~DispatchGate();
    DispatchGate(const DispatchGate &) = delete;
    DispatchGate &operator=(const DispatchGate &) = delete;
    GateMark mark() const { return m_; }

  private:
    GateMark m_;
    void *gate_ = nullptr;
};

// how often a launch found the one before it (another stream's) not yet fully dispatched, and how often it gave up waiting (tests, tools)
void dispatch_gate_stats(unsigned long long *waits, unsigned long long *timeouts);

} // namespace pixo_dev
