// jpeg_scan_dev.h — device helpers shared by the single-pass entropy kernels (jpeg_scan_fused.hip: scan_code, prog_code,
// stuff_fused) and the fused pixel -> bit stream kernel (jpeg_pixels_code.hip): the geometry of a group, wavefront scans,
// agent-scope descriptor accesses, the bounded decoupled look-backs, the LDS sinks of the flat walk.
#pragma once
#include <hip/hip_runtime.h>

#include "jpeg_scan_block.h"

namespace pixo_dev {
using namespace pixo_scan;
namespace {
constexpr int kGroup = 192;                  // lanes = blocks per group
constexpr int kGroupWaves = kGroup / 64;
#ifndef PIXO_WINDOW_WORDS
#define PIXO_WINDOW_WORDS 1536
#endif
#ifndef PIXO_SCRATCH_WORDS
#define PIXO_SCRATCH_WORDS 12
#endif
constexpr uint32_t kWindowWords = PIXO_WINDOW_WORDS;  // the LDS bit buffer: 6 KiB, one round for a group of noise at q = 80 (5.4 KiB); a longer group takes several
constexpr uint32_t kBufWords = kWindowWords + kGroup; // + one dummy word per lane for the sinks.  (19 KiB of LDS per group in all: eight groups per CU.)
constexpr uint32_t kScratchWords = PIXO_SCRATCH_WORDS; // per lane: a block of up to 384 bits is coded in ONE walk (noise at q = 80: 230 +- 30)
constexpr uint32_t kScratchPitch = kScratchWords + 1; // + the dummy word; odd: lane-strided accesses hit all banks
constexpr uint64_t kFlagAggregate = 1ull << 62, kFlagPrefix = 2ull << 62, kValueMask = (1ull << 62) - 1;
constexpr uint64_t kTailValid = 1ull << 63;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// ---- wavefront inclusive scan: four row shifts inside the rows of 16, two row broadcasts across them ----------
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_add(uint32_t v)
{
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    v = dpp_add<0x111, 0xF>(v); // row_shr:1
    v = dpp_add<0x112, 0xF>(v); // row_shr:2
    v = dpp_add<0x114, 0xF>(v); // row_shr:4
    v = dpp_add<0x118, 0xF>(v); // row_shr:8
    v = dpp_add<0x142, 0xA>(v); // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v); // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ unsigned long long load_relaxed(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_relaxed(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum of everything before ticket `g` (decoupled look-back), by one whole wavefront: per round lane l inspects the
// descriptors of tickets top - l - 64 i, i < kLookBatch (all loads in flight together); a descriptor holds flag + value
// in ONE 64-bit word, so a single relaxed load sees a consistent pair.  Every lane returns the sum.
// Round width: every group of a launch is resident and reaches this point at about the same time, and a descriptor
// access is a round trip to the fabric (the XCDs' L2s are not coherent with each other).  One LANE walking back took
// 100 us for the 4096x4096 image (round 3).  Rounds of 512 were the round-3 choice; with the round-4 kernels (tails
// published by every group, aggregate tails) the A/B of 64 / 128 / 256 / 512 / 1024 / 2048 per round has 64 equal on
// baseline files and 7-11 us faster on progressive ones, and the wide rounds clearly slower (1024: +30..+70 us) — the
// descriptor reads of 2048+ simultaneous look-backs are the traffic that matters.
#ifndef PIXO_LOOK_BATCH
#define PIXO_LOOK_BATCH 1 // 64 predecessors per round; 2..32 measured slower (profiles/r04_ab_look_batch.txt)
#endif
constexpr int kLookBatch = PIXO_LOOK_BATCH;
// what a polling wavefront sleeps between two reads of a descriptor that is not there yet (units of 64 cycles)
#ifndef PIXO_POLL_SLEEP
#define PIXO_POLL_SLEEP 1
#endif
constexpr int kPollSleep = PIXO_POLL_SLEEP;
// Waiting is BOUNDED: forward progress of these kernels rests on the hardware starting the workgroups of a
// grid in increasing id order (file header) — observed, not promised by HIP.  Every poll loop gives up after `budget`
// polls (launch argument; 2^20 polls of >= 64 cycles each = tens of milliseconds, three orders of magnitude beyond any
// real wait) and raises the launch's abort flag, which every other waiting workgroup checks as well: the kernel then
// ends with garbage in its outputs instead of hanging the GPU, and the host — which reads the flag from the pinned
// mailbox behind the stream's synchronisation — codes the scan again with the multi-pass kernels of jpeg_entropy.hip.
// (one lane) raises the launch's abort flag, on the device and in the pinned mailbox
__device__ __forceinline__ void raise_abort(unsigned long long *abort_flag, unsigned long long *host_abort)
{
    __hip_atomic_store(abort_flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (host_abort) *host_abort = 1ull;
}
constexpr uint64_t kLookBackFailed = ~0ull;
__device__ __forceinline__ void publish_aggregate(unsigned long long *desc, uint64_t g, uint64_t floor, uint64_t aggregate)
{ // (one lane) as early as possible: the groups behind this one wait for it.  `floor`: first ticket of g's chain (0; the
  // first group of g's segment): it has nothing before it, its aggregate IS its inclusive prefix
    store_relaxed(&desc[g], (g == floor ? kFlagPrefix : kFlagAggregate) | aggregate);
}
__device__ __forceinline__ uint64_t look_back(unsigned long long *desc, uint64_t g, uint64_t floor, uint64_t aggregate, unsigned long long *abort_flag,
                                              unsigned long long *host_abort, uint32_t budget)
{ // (after publish_aggregate); kLookBackFailed: gave up waiting
    uint32_t polls = 0;
    const int lane = threadIdx.x & 63;
    if (g == floor) return 0;
    uint64_t before = 0;
    for (int64_t top = (int64_t)g - 1;; top -= 64 * kLookBatch) {
        unsigned long long d[kLookBatch];
#pragma unroll
        for (int i = 0; i < kLookBatch; i++) {
            const int64_t j = top - lane - 64 * i;
            d[i] = j >= (int64_t)floor ? load_relaxed(&desc[j]) : kFlagPrefix; // (below the chain's first ticket: an inclusive prefix of nothing)
        }
        bool done = false;
#pragma unroll
        for (int i = 0; i < kLookBatch; i++) {
            if (done) break; // (wave-uniform)
            const int64_t j = top - lane - 64 * i;
            bool gave_up = false;
            while ((d[i] >> 62) == 0) {
                __builtin_amdgcn_s_sleep(kPollSleep);
                if (++polls > budget) { gave_up = true; break; }
                d[i] = load_relaxed(&desc[j]);
            }
            if (PIXO_ANY64(gave_up)) { // (wave-uniform)
                if (gave_up) raise_abort(abort_flag, host_abort);
                return kLookBackFailed;
            }
            const uint64_t have_prefix = __builtin_amdgcn_ballot_w64((d[i] >> 62) == 2);
            const int first = have_prefix ? __builtin_ctzll(have_prefix) : 64; // nearest predecessor that knows its inclusive prefix
            // aggregates of the lanes in front of it (< 2^19 each: 32-bit sum), plus its prefix
            const uint32_t part = lane < first ? (uint32_t)(d[i] & kValueMask) : 0u;
            before += (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(part), 63);
            if (have_prefix) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(d[i] & 0xFFFFFFFFu), first);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((d[i] & kValueMask) >> 32), first);
                before += ((uint64_t)hi << 32) | lo;
                done = true;
            }
        }
        if (done) break;
    }
    return before; // (the caller publishes before + aggregate as this ticket's inclusive prefix)
}

// The same sum by REDUCE-THEN-SCAN in two levels (round 4): nothing but aggregates is ever read, so no group waits for
// another group's look-back.  Groups are taken in blocks of 64 (counted from the chain's floor); the LAST group of a block
// adds up its block — its own aggregate and the 63 before it, one round — and publishes the block's sum in `sup`; a group
// then needs the aggregates in front of it inside its own block (<= 63, one load per lane) and the sums of all blocks
// before (one load per lane and 4096 groups), all issued together.  Critical path: aggregates -> block sums -> done, two
// fabric round trips, where the chained form above takes up to g / 64 rounds (64 predecessors per round) when all groups of a launch arrive at once (a
// smooth image: 4.5 us median, 8 us for the last groups of 2048; profiles/r04_scan_code_timeline.txt).
// desc[g] must hold kFlagAggregate | aggregate (publish_aggregate also writes kFlagPrefix for the floor: any flag counts).
// `sup`: the chain's block sums (zero before the launch).  Every lane returns the sum; kLookBackFailed: gave up waiting.
// How many copies of its block sums a launch keeps and how far apart (u64 words): launches of 256 groups and more keep 16 copies,
// each 4 KiB + 256 B behind the one before (or the sums' own size rounded up to 256 B, + 256 B): different memory channels.
struct SupLayout { uint32_t copies, stride; };
__host__ __device__ inline SupLayout sup_layout(uint64_t groups, uint64_t sums)
{
    if (groups < 256) return SupLayout{1u, (uint32_t)sums};
    return SupLayout{16u, (uint32_t)(sums <= 512 ? 512 + 32 : ((sums + 31) / 32) * 32 + 32)};
}
// COPIES of the block sums (round 6): every group of a launch reads ALL block sums before it — 2048 groups polling the same two
// cache lines, which one memory channel serves one request after the other (agent-scope loads are not cached in the XCDs' L2s):
// the look-back's time followed the number of descriptor bytes, not the number of round trips.  The block's last group therefore
// stores its sum `copies` times, `stride` words apart (different channels), and a group reads the copy its ticket selects.
__device__ __forceinline__ uint64_t look_back_blocks(unsigned long long *desc, unsigned long long *sup, uint64_t g, uint64_t floor, uint64_t aggregate,
                                                     unsigned long long *abort_flag, unsigned long long *host_abort, uint32_t budget,
                                                     uint32_t copies = 1, uint32_t stride = 0)
{
    // (the lane as a value of THIS call: the two 64-bit addresses a lane reads from are then computed here — as expressions of the
    // kernel's lane index they are loop invariants of whatever encloses the call, get hoisted, live across the callers' walks and
    // end in scratch memory: 7 of the 8 spilled dwords of pixels_code_kernel, 11 MB of scratch writes a launch)
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const uint64_t rel = g - floor, k = rel >> 6;
    const uint32_t in_block = (uint32_t)(rel & 63);
    const uint64_t block_first = floor + (k << 6);
    uint32_t polls = 0;
    bool gave_up = false;
    unsigned long long *const sup_w = sup;                  // copy 0 (the writer's base)
    sup += (size_t)((uint32_t)g & (copies - 1u)) * stride;  // the copy this group reads
    // (A) the aggregates in front of g inside its block, (B) the first 64 block sums — in flight together
    unsigned long long a = (uint32_t)lane < in_block ? load_relaxed(&desc[block_first + lane]) : kFlagAggregate;
    unsigned long long b = (uint64_t)lane < k ? load_relaxed(&sup[lane]) : kFlagAggregate;
    while ((a >> 62) == 0 && !gave_up) {
        __builtin_amdgcn_s_sleep(kPollSleep);
        if (++polls > budget) gave_up = true; else a = load_relaxed(&desc[block_first + lane]);
    }
    if (PIXO_ANY64(gave_up)) {
        if (gave_up) raise_abort(abort_flag, host_abort);
        return kLookBackFailed;
    }
    const uint32_t in_front = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan((uint32_t)(a & kValueMask)), 63); // (< 64 x 2^19)
    if (in_block == 63 && (uint32_t)lane < copies) store_relaxed(&sup_w[(size_t)lane * stride + k], kFlagAggregate | ((uint64_t)in_front + aggregate)); // this block's sum, before waiting for the others'
    uint64_t before = in_front;
    for (uint64_t base = 0;;) { // block sums, 64 per round (one round up to 4096 groups)
        while ((b >> 62) == 0 && !gave_up) {
            __builtin_amdgcn_s_sleep(kPollSleep);
            if (++polls > budget) gave_up = true; else b = load_relaxed(&sup[base + lane]);
        }
        if (PIXO_ANY64(gave_up)) {
            if (gave_up) raise_abort(abort_flag, host_abort);
            return kLookBackFailed;
        }
        // (a block sum is below 64 x 2^19 = 2^25: 64 of them fit 32 bits)
        before += (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan((uint32_t)(b & kValueMask)), 63);
        base += 64;
        if (base >= k) break;
        b = base + lane < k ? load_relaxed(&sup[base + lane]) : kFlagAggregate;
    }
    return before;
}

// (one lane of every workgroup, at its start) dispatch_gate.hpp: the launch's last eight workgroups — one per XCD — say that they have
// started; with fewer than eight workgroups the first one speaks for the missing ones.  lin: the workgroup's linear id, total: all of them.
__device__ __forceinline__ void dispatch_mark(unsigned long long *slots, unsigned long long seq, uint64_t lin, uint64_t total)
{
    if (!slots) return;
    const uint64_t j = total - 1u - lin;
    if (j < 8u) __hip_atomic_store(&slots[j], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lin == 0 && total < 8u)
        for (uint64_t k = total; k < 8u; k++) __hip_atomic_store(&slots[k], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- the LDS bit buffer: sink of block_pack_flat (jpeg_scan_block.h) over a window of words --------------------------
// Word i of the window is word `first + i` of the stream; every word starts out zero and is only ever OR-ed (LDS
// atomic without return).  Words outside [0, limit) — a group of very long blocks is written out in more than one
// round — and the "no flush" case go to a per-lane dummy word behind the window: no branch.
// ---- the per-lane scratch of the single walk: a block's bits from bit 0, words stored plainly ------------------------
// (the word being filled is stored after every symbol — the last store, the complete word, wins: cheaper than a
// select on "complete"; words beyond the scratch — a block of more than kScratchWords * 32 bits — go to the lane's
// dummy word)
struct LaneSink {
    uint32_t *words; // this lane's kScratchWords words + 1 dummy
    __device__ __forceinline__ void or_word(bool, uint32_t word, uint32_t value)
    {
        words[word < kScratchWords ? word : kScratchWords] = value;
    }
};

struct LdsSink {
    uint32_t *buf;
    uint32_t limit, dummy;
    __device__ __forceinline__ void or_word(bool flush, uint32_t word, uint32_t value)
    {
        const uint32_t i = (flush && word < limit) ? word : dummy;
        (void)__hip_atomic_fetch_or(&buf[i], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};
__device__ __forceinline__ uint32_t zero_byte_mask(uint32_t x)
{ // 0x80 in every byte of x that is zero (exact)
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}

} // namespace
} // namespace pixo_dev
