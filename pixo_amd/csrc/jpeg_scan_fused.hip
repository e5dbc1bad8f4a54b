// jpeg_scan_fused.hip — the device entropy stage of one uninterrupted baseline scan in TWO kernels, each a
// single pass over its input (round 1 needed three passes over the coefficient tuple, two
// prefix-sum launches, a memset and two host round trips — jpeg_entropy.hip, which still serves scans with
// restart markers, batches and progressive scans):
//
//   code   tuple -> packed MSB-first bit stream.  A group of 192 lanes owns 192 consecutive blocks of the scan
//          order (32 MCUs of 4:2:0, 64 of 4:4:4): every lane loads its block into 32 registers and walks it ONCE
//          (encode_block, src/jpeg/huffman.rs:423-481; the branch-free block_pack_flat of jpeg_scan_block.h), coding it
//          from bit 0 into a few words of LDS of its own — where the packer stands at the end is the block's length;
//          the lengths are summed with wavefront scans (DPP row shifts + row broadcasts, no LDS), every lane moves
//          its words, shifted, into the group's LDS bit buffer at its group-relative offset, the group's position in
//          the stream comes from a decoupled look-back over the groups before it (one 64-bit descriptor per group),
//          and the group writes the buffer out with coalesced dword stores.  The word two neighbouring groups share is
//          written by the later one, which receives the earlier one's bits through a second descriptor: no atomics
//          on the stream, no zero-filling of it.  A scan can be coded in pieces (ScanPiece) that hand each other the
//          bit position on the device.
//   stuff  packed stream -> final bytes with 0x00 after every 0xFF (BitWriterMsb, src/bits.rs:245-253), 16 KiB
//          tiles: 0xFF census per word, wavefront scans, look-back for the tile's output position, bytes expanded
//          into LDS and written out as aligned 16-byte stores.
//   count  (optimised tables) the same walk, counting symbols instead of coding them; per-workgroup rows summed by
//          a second small kernel.
//
// Neither kernel needs a host round trip: `stuff` reads the stream's length where `code` left it.
//
// Forward progress.  A group / tile waits only for LOWER workgroup ids (look-back, hand-off of the shared word).  The
// hardware hands the workgroups of a 1-D grid to each XCD in increasing id order, so the smallest unfinished id is
// either running or the next one its XCD starts: nothing it waits for can be missing.  (Taking ids from an atomic
// ticket counter would not need that property — it was tried first: 2 x 2048 device-scope atomics on one address
// cost 70 us per launch, more than the rest of the kernel.  Workgroups that stride over several tiles would need all
// of them resident at once, which two such kernels from two host threads can deny each other.)  `stuff` does not know
// the number of tiles when it is launched: it gets one workgroup per tile of a generous guess — surplus workgroups
// leave at once — and, if the stream turns out longer, a second launch for the tiles behind the guess.
#include <hip/hip_runtime.h>

#include "dispatch_gate.hpp"
#include "jpeg_entropy.hpp"
#include "jpeg_scan_block.h"
#include "jpeg_scan_dev.h"

namespace pixo_dev {
using namespace pixo_scan;

namespace {

// SEG (segmented scans: the images of a batch, restart intervals — SegArgs in jpeg_entropy.hpp): the scan consists of
// byte-aligned segments of seg.blocks blocks, each coded from DC predictors 0 into a packed stream of ITS OWN (region
// seg.stream_words * segment of `stream`) and padded with 1-bits like the end of a scan; the groups of 192 blocks are
// aligned with the segments (the last group of a segment is partly empty), a segment's first group is the floor of its
// groups' look-back — nothing crosses a segment — and its last group leaves the segment's length in seg.bits.  How the
// segments follow each other in the file is the stuffing kernel's business.
#ifdef PIXO_TIMELINE // (experiment builds only, tools/scan_timeline.py: where a group's time goes; 100 MHz constant clock)
__device__ unsigned long long g_timeline[8192 * 8];
#define PIXO_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_timeline[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PIXO_STAMP(k) do { } while (0)
#endif
template <int MODE, bool SEG>
__global__ __launch_bounds__(kGroup) __attribute__((amdgpu_waves_per_eu(6, 6))) void scan_code_kernel
(const ScanArgs a, unsigned long long *state, uint32_t *stream, unsigned long long *clear, uint32_t clear_words,
 unsigned long long *host_totals, const ScanPiece piece, const SegArgs seg, uint32_t spin_budget, const GateMark gate)
{
    if (threadIdx.x == 0) dispatch_mark(gate.slots, gate.seq, blockIdx.x, gridDim.x);
    // state: [0] abort flag, [1] total bits (out), [2 ..] per group: descriptor, then per group: tail
    __shared__ __attribute__((aligned(16))) uint32_t buf[kBufWords];
    __shared__ uint32_t tab[kWalkWords]; // the Huffman tables in the flat walk's form (jpeg_scan_block.h)
    __shared__ uint32_t scratch[kGroup * kScratchPitch];
    __shared__ uint32_t wave_sum[kGroupWaves], wave_long[kGroupWaves];
    __shared__ unsigned long long s_before;
    __shared__ uint32_t s_carry, s_abort;
    const int lane = threadIdx.x, wave = lane >> 6;
    PIXO_STAMP(0);
    if (lane == 0) { s_carry = 0; s_abort = 0; }
    const uint64_t g = blockIdx.x; // (one group per workgroup; see the note on dispatch order at the top of the file)
    // the group's place: segment, first ticket of the segment (the look-back's floor), blocks
    uint64_t floor_g = 0, first_in_chain = g * kGroup, nblocks_chain = a.nblocks, sidx = 0;
    if (SEG) {
        sidx = g / seg.groups;
        floor_g = sidx * seg.groups;
        const uint64_t seg_first_block = sidx * seg.blocks;
        nblocks_chain = a.nblocks - seg_first_block < seg.blocks ? a.nblocks - seg_first_block : seg.blocks;
        first_in_chain = (g - floor_g) * kGroup;
        stream += sidx * seg.stream_words;
    }
    const bool surplus = SEG && first_in_chain >= nblocks_chain; // (the last segment is shorter: its surplus groups only do the housekeeping)
    const uint64_t ngroups_total = SEG ? seg.nsegs * seg.groups : (a.nblocks + kGroup - 1) / kGroup;
    unsigned long long *desc = state + 2, *tails = state + 2 + ngroups_total;
    // block sums of the two-level look-back: per chain (the scan; a segment) one word per 64 groups, behind the tails — in copies
    // (sup_layout, jpeg_scan_dev.h: this is copy 0)
    unsigned long long *sup = state + 2 + 2 * ngroups_total + (SEG ? sidx * ((seg.groups + 63) >> 6) : 0);
    const SupLayout sl = sup_layout(ngroups_total, SEG ? seg.nsegs * ((seg.groups + 63) >> 6) + 1 : ((ngroups_total + 63) >> 6) + 1);
    unsigned long long *const host_abort = host_totals ? host_totals + 3 : nullptr;
    // A scan coded piece by piece (ScanPiece, jpeg_entropy.hpp): the piece's stream is byte-aligned with the SCAN — its
    // first `lead` bits are the end of the piece before — so that the stuffing kernel can work on it without a shift.
    const uint64_t bits_before_piece = piece.index ? piece.chain[piece.index] : 0ull;
    const uint32_t lead = (uint32_t)(bits_before_piece & 7);
    if (piece.chain && piece.index == 0 && blockIdx.x == 0 && lane == 0) piece.chain[0] = 0; // (read by piece 1)
    // ---- blocks in, before anything else (the barrier below then waits once for these, the tables and the housekeeping):
    // 8 x 16 bytes per lane straight into 32 registers.  A lane's block is one 128-byte line that its eight loads touch
    // one after the other: cached loads (the line stays in L1 for the other seven), not non-temporal ones.  (Staging
    // the group through LDS for perfectly coalesced loads cost 24 KiB per group and 60 more VGPRs for the addresses:
    // half the occupancy.)
    const bool live = !surplus && first_in_chain + lane < nblocks_chain;
    const uint64_t s = piece.first_block + (SEG ? sidx * seg.blocks : 0) + first_in_chain + lane;
    uint32_t w[32];
    // DC predictor: the previous block of the same component (jpeg/mod.rs:1417-1419); the scan's first blocks start
    // from the seed (0, or the DCs above a band); a segment's first blocks from 0 (jpeg/mod.rs:1441-1444)
    int prev_dc = 0, cls = 0;
    {
        const BlockRef ref = block_of(MODE, live ? s : 0);
        const int16_t *base = ref.comp == 0 ? a.y : (ref.comp == 1 ? a.cb : a.cr);
        const v4u *p = reinterpret_cast<const v4u *>(base + ref.index * 64);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const v4u q = p[r];
            w[4 * r] = q.x; w[4 * r + 1] = q.y; w[4 * r + 2] = q.z; w[4 * r + 3] = q.w;
        }
        if (live) {
            prev_dc = ref.index ? (int)base[(ref.index - 1) * 64] : (int)a.seed_dc[ref.comp];
            cls = ref.comp == 0 ? 0 : 1;
            if (SEG) { // the first block of every component in the segment's first MCU
                const uint64_t in_seg = first_in_chain + lane;
                const bool first_of_comp = MODE == 2 ? (in_seg == 0 || in_seg == 4 || in_seg == 5) : in_seg < (uint64_t)a.blocks_per_mcu;
                if (first_of_comp) prev_dc = 0;
            }
        }
    }
    for (int i = lane; i < kWalkWords; i += kGroup) tab[i] = a.tables[kTableWords + i]; // (the walk's form lies behind the packed one)
    // (housekeeping for the kernel that follows: its descriptors must be zero when it starts — cheaper here than a memset launch)
    for (uint64_t i = (uint64_t)blockIdx.x * kGroup + lane; i < clear_words; i += (uint64_t)gridDim.x * kGroup) clear[i] = 0;
    if (surplus) return;
    __syncthreads();
    PIXO_STAMP(1);
    {
        // ---- THE walk: the block's codes into the lane's scratch from bit 0 — which also gives its length; group scan;
        // the group's aggregate goes out at once.  Blocks of more than 384 bits do not fit the scratch: a group that
        // holds one is packed by a second walk below (noise at q >= 90, not photographs).
        uint32_t len;
        {
            FlatPack<LaneSink> p;
            p.sink = LaneSink{scratch + lane * kScratchPitch};
            p.acc = 0; p.pending = 0; p.word = 0;
            block_pack_flat(w, prev_dc, tab + cls * kWalkClassWords, p);
            len = p.word * 32u + p.pending;
            p.finish();
        }
        if (!live) len = 0;
        const bool long_block = len > kScratchWords * 32u;
        const uint32_t incl = wave_inclusive_scan(len);
        if ((lane & 63) == 63) wave_sum[wave] = incl;
        const bool any_long = PIXO_ANY64(long_block); // (a ballot: outside the one-lane branch below)
        if ((lane & 63) == 0) wave_long[wave] = any_long ? 1u : 0u;
        __syncthreads();
        PIXO_STAMP(2);
        uint32_t wave_base = 0, group_bits = 0, group_long = 0;
#pragma unroll
        for (int k = 0; k < kGroupWaves; k++) {
            if (k < wave) wave_base += wave_sum[k];
            group_bits += wave_sum[k];
            group_long |= wave_long[k];
        }
        if (lane == 0) publish_aggregate(desc, g, floor_g, group_bits);
        const bool last_group = first_in_chain + kGroup >= nblocks_chain;
        // ---- the blocks' bits at GROUP-RELATIVE offsets into the LDS buffer — the position in the stream is not needed
        // for that, and meanwhile the aggregates of the groups before travel — one window of kWindowWords words per
        // round (usually one): gathered from the scratches, or (a group with a long block) packed by a second walk;
        // then the look-back, then the write-out, shifted.
        const uint32_t my_bit = wave_base + (incl - len);
        const uint32_t local_words = (group_bits + 31) >> 5; // >= 1: every block has bits
        // (known after the look-back of the first round)
        uint64_t first_word = 0;
        uint32_t sh = 0, out_words = 0, pad_word = ~0u, pad_mask = 0;
        bool tail_partial = false;
        uint32_t head_word = 0; // (lane 0) this group's bits of the stream word it shares with the group before
        for (uint32_t wbase = 0; wbase < local_words; wbase += kWindowWords) {
            const uint32_t wn = local_words - wbase < kWindowWords ? local_words - wbase : kWindowWords;
            for (uint32_t i = lane; i < wn; i += kGroup) buf[i] = 0;
            __syncthreads();
            const int64_t rel = (int64_t)my_bit - (int64_t)wbase * 32;
            if (!group_long) { // every word of the lane's scratch, shifted to its place (two LDS ORs per word)
                const uint32_t nw = (len + 31) >> 5, bsh = (uint32_t)(rel & 31), dummy = kWindowWords + (uint32_t)lane;
                const uint32_t d0 = (uint32_t)(rel >> 5); // wraps below zero for words before the window
#pragma unroll
                for (uint32_t j = 0; j < kScratchWords; j++) {
                    if (!PIXO_ANY64(j < nw)) break; // (wave-uniform)
                    const uint32_t v = j < nw ? scratch[lane * kScratchPitch + j] : 0u;
                    const uint32_t d = d0 + j;
                    (void)__hip_atomic_fetch_or(&buf[d < wn ? d : dummy], v >> bsh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    (void)__hip_atomic_fetch_or(&buf[d + 1 < wn ? d + 1 : dummy], bsh ? v << (32 - bsh) : 0u, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            // (opaque to the optimiser: otherwise everything the first walk derived from the 63 coefficients — values,
            // magnitude categories, table words — stays alive for the second walk: 245 VGPRs instead of ~70)
#pragma unroll
            for (int i = 0; i < 32; i++) asm volatile("" : "+v"(w[i]));
            if (group_long && PIXO_ANY64(live && rel < (int64_t)wn * 32 && rel + (int64_t)len > 0)) { // (some block of the wavefront lies in the window)
                FlatPack<LdsSink> p;
                p.sink = LdsSink{buf, live ? wn : 0u, kWindowWords + (uint32_t)lane};
                p.acc = 0;
                p.pending = (uint32_t)(rel & 31);
                p.word = (uint32_t)(rel >> 5); // relative to the window; wraps below zero for words before it
                block_pack_flat(w, prev_dc, tab + cls * kWalkClassWords, p);
                p.finish();
            }
            if (wbase == 0) { // where the group starts in the stream
                PIXO_STAMP(3);
                if (wave == 0) {
#ifdef PIXO_LOOK_CHAINED // (A/B: the chained decoupled look-back of rounds 2-4)
                    const uint64_t sum = look_back(desc, g, floor_g, group_bits, state, host_abort, spin_budget);
#else
                    const uint64_t sum = look_back_blocks(desc, sup, g, floor_g, group_bits, state, host_abort, spin_budget, sl.copies, sl.stride);
#endif
                    if (lane == 0) {
                        if (sum == kLookBackFailed) s_abort = 1;
                        s_before = sum;
#ifdef PIXO_LOOK_CHAINED
                        if (g != floor_g) store_relaxed(&desc[g], kFlagPrefix | (sum + group_bits));
#endif
                        if (last_group) { // the stream's length in bits (unpadded; a later piece: with its leading bits)
                            if (SEG) {
                                seg.bits[sidx] = sum + group_bits;
                            } else {
                                state[1] = lead + sum + group_bits;
                                if (host_totals) host_totals[0] = lead + sum + group_bits;
                                if (piece.chain) piece.chain[piece.index + 1] = bits_before_piece + sum + group_bits;
                            }
                        }
                    }
                }
                __syncthreads();
                PIXO_STAMP(4);
                if (s_abort) return; // (the look-back gave up: the host codes this scan again, see Waiter)
                const uint64_t start = lead + s_before;
                uint64_t end = start + group_bits;
                if (last_group && a.pad_last) { // BitWriterMsb::flush pads the last byte with 1-bits
                    const uint32_t n = (uint32_t)((8 - (end & 7)) & 7);
                    if (n) {
                        pad_word = (uint32_t)((end >> 5) - (start >> 5));
                        pad_mask = ((1u << n) - 1u) << (32 - (uint32_t)(end & 31) - n);
                    }
                    end += n;
                }
                first_word = start >> 5;
                sh = (uint32_t)(start & 31);
                out_words = (uint32_t)((end - (first_word << 5) + 31) >> 5); // local_words or local_words + 1
                tail_partial = (end & 31) != 0 && !last_group;                 // the last word is finished by a later group
            } else {
                __syncthreads();
            }
            // ---- out.  Stream word first_word + j = the buffer's words j - 1 and j funnelled by `sh`; the group writes
            // every word it completes, except the word it shares with the group before (j = 0 when sh != 0): that one
            // waits for the other group's bits.  The word it leaves unfinished goes to the next group as `tail`.
            const uint32_t carry = s_carry; // the previous window's last word
            const bool last_round = wbase + wn == local_words;
            const uint32_t upto = last_round ? out_words - wbase : wn; // words of this round (the last round may have one more)
            uint32_t word0 = 0, tail_word = 0;
            for (uint32_t i = lane; i < upto; i += kGroup) {
                const uint32_t j = wbase + i;
                const uint32_t cur = i < wn ? buf[i] : 0u, prev = i ? buf[i - 1] : carry;
                uint32_t v = sh ? (cur >> sh) | (prev << (32 - sh)) : cur;
                v |= j == pad_word ? pad_mask : 0u;
                const bool is_head = j == 0 && sh != 0, is_tail = tail_partial && j + 1 == out_words;
                if (is_head) word0 = v;
                if (is_tail) tail_word = v;
                if (!is_head && !is_tail) __builtin_nontemporal_store(v, &stream[first_word + j]);
            }
            // (j = 0 is lane 0's in the first round; the tail word belongs to lane (upto - 1) % kGroup of the last round)
            const bool has_tail = last_round && tail_partial;
            const bool pass_through = has_tail && out_words == 1 && sh != 0; // (a handful of bits inside one word)
            if (has_tail && !pass_through && (uint32_t)lane == (upto - 1) % kGroup) store_relaxed(&tails[g], kTailValid | tail_word);
            if (wbase == 0) head_word = word0;
            if (lane == 0) s_carry = buf[wn - 1];
            __syncthreads();
        }
        PIXO_STAMP(5);
        // ---- the word shared with the group before: its bits arrive as that group's tail.  AFTER this group's own
        // tail went out: waiting here in the first round of several made a chain through every group of the scan (each
        // link one round: 310 us for 2048 groups of two rounds, against 45).
        if (sh != 0 && lane == 0) {
            uint32_t inherited = 0;
            if (g > floor_g) {
                unsigned long long t = load_relaxed(&tails[g - 1]);
                uint32_t polls = 0;
                while (!(t & kTailValid)) {
                    __builtin_amdgcn_s_sleep(kPollSleep);
                    if (++polls > spin_budget) { raise_abort(state, host_abort); return; }
                    t = load_relaxed(&tails[g - 1]);
                }
                inherited = (uint32_t)t;
            } else if (piece.index) { // the `lead` bits of the byte shared with the piece before: the last, partial byte of its stream
                const uint64_t before_prev = piece.chain[piece.index - 1];
                const uint64_t prev_bits = (before_prev & 7) + (bits_before_piece - before_prev);
                const uint64_t at = prev_bits >> 3; // (byte index in the previous stream)
                const uint32_t byte = (piece.prev_stream[at >> 2] >> (24 - 8 * (uint32_t)(at & 3))) & 0xFFu;
                inherited = (byte & (0xFF00u >> lead) & 0xFFu) << 24;
            }
            const uint32_t merged = inherited | head_word;
            if (tail_partial && out_words == 1) store_relaxed(&tails[g], kTailValid | merged); // (pass-through: a handful of bits inside one word)
            else __builtin_nontemporal_store(merged, &stream[first_word]);
        }
        PIXO_STAMP(6);
    }
}
#ifdef PIXO_TIMELINE
extern "C" __attribute__((visibility("default"))) int pixo_hip_debug_scan_timeline(unsigned long long *out, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

// ---- progressive scans in one pass (round 4) ----------------------------------------------------------------------------
// The scans of simple_progressive_script (progressive.rs:98-110; jpeg/mod.rs:872-927, :1248-1380) as segments of ONE launch:
// group g belongs to scan k when first_group[k] <= g < first_group[k + 1]; lane l of it codes block (g - first_group[k]) * 192
// + l of the scan's component (storage order) — in a DC scan four consecutive blocks, 768 per group.  What differs from scan_code_kernel<SEG>:
//   * segments of different sizes, each with its own walk: DC scans code one symbol per block (encode_dc_first), AC scans the
//     band [ss, se] without an end-of-block code (band_pack_flat);
//   * the END-OF-BAND RUN counter (progressive.rs:156-162, :171-174, :206-209, :313-345) crosses blocks: what it makes a lane
//     emit around its own symbols (band_edge, jpeg_scan_block.h) depends on C_b, the count of run contributions since the last
//     non-empty block — inside the wavefront two ballots, across wavefronts three LDS words, across groups a look-back of
//     its own (descriptor A: a group that holds a non-empty block publishes the count behind it as a PREFIX at once; a
//     group of empty blocks publishes its count as an aggregate and, once it knows what flows in, the sum as a prefix).  The
//     A descriptors go out right after the walk, before any length is known, so that look-back rarely waits;
//   * a group may have no bits at all (192 empty blocks): it leaves an aggregate of zero bits in look-back B and a
//     "transparent" mark in its tail slot (the stream word two groups share is found by a look-back over the tails) and is done.
//
// Round 5: ONE LOAD and ONE WALK of a block serve every scan it takes part in.  A workgroup holds 192 consecutive blocks of ONE
// component: a lane reads its block once, codes its DC symbol (dc_symbol_bits) and walks its 63 AC positions once
// (bands_pack_flat: luminance as the bands 1..10 and 11..63 back to back in the lane's scratch, chrominance as 1..63) — and then
// is a link in the chains of up to THREE scans at the same place (its component's DC scan and AC scans): three bit counts and
// two run counters published together, the look-backs of the three scans run side by side on the group's three wavefronts,
// and the scans' bits are placed and written out one after the other through the same LDS window.  (Round 4 gave every scan
// workgroups of its own: a luminance block was read three times — 110-134 MB of HBM traffic for a 50 MB tuple — and walked
// twice; a first round-5 version that kept one workgroup per component but ran the scans as passes, one after the other, paid
// the look-backs' latency once per pass: 85 us where round 4 took 75, the DC pass alone 20 us for 2 % of the bits.)
// Every wait still points at a lower workgroup id.  The look-back over the bit counts is the reduce-then-scan form in two
// levels (look_back_blocks) like scan_code's; the run counter's and the shared word's keep the chained form (their sums are
// not sums).
// state: [0] abort flag, [1] -, then per (scan, group): descriptor A, descriptor B, tail; then the block sums of the B chains.
constexpr uint32_t kEobSyms = 16; // the packed words of the class's symbols 0x00 .. 0xE0 (end-of-band runs), one spare
__global__ __launch_bounds__(kGroup) __attribute__((amdgpu_waves_per_eu(6, 6))) void prog_code_kernel
(const ProgCode a, const SegArgs seg, unsigned long long *state, uint32_t *stream0, unsigned long long *clear, uint32_t clear_words,
 unsigned long long *host_totals, uint32_t spin_budget, const GateMark gate)
{
    if (threadIdx.x == 0) dispatch_mark(gate.slots, gate.seq, blockIdx.x, gridDim.x);
    __shared__ __attribute__((aligned(16))) uint32_t buf[kBufWords];
    __shared__ uint32_t tab[kWalkClassWords];
    __shared__ uint32_t eobs[kEobSyms];
    __shared__ uint32_t scratch[kGroup * kScratchPitch];
    __shared__ uint32_t wave_sum[3][kGroupWaves], wave_long[kGroupWaves], wave_any[2][kGroupWaves], wave_tail[2][kGroupWaves];
    __shared__ int16_t s_dc[kGroup], s_ext_dc;
    __shared__ unsigned long long s_before[3], s_in[2];
    __shared__ uint32_t s_carry, s_abort, s_head[3], s_share[3];
    __shared__ unsigned long long s_first_word[3];
    if (threadIdx.x == 0) s_abort = 0;
    if (threadIdx.x < 3) s_share[threadIdx.x] = 0;
    PIXO_STAMP(0);
    uint32_t slot = 0; // which run of workgroups this one belongs to: comp_first[slot] <= id < comp_first[slot + 1]
#pragma unroll
    for (uint32_t i = 1; i < 3; i++) slot += blockIdx.x >= a.comp_first[i] ? 1u : 0u;
    const uint32_t comp = a.comp_order[slot]; // the component this workgroup's blocks belong to (uniform)
    const uint64_t local = blockIdx.x - a.comp_first[slot]; // the group's place among the component's groups
    const uint64_t nblocks_chain = a.comp_blocks[comp];
    const uint32_t nseg = a.comp_npass[comp]; // scans of the component: 2 (DC, 1..63) or 3 (DC, 1..10, 11..63)
    const bool split = nseg == 3;
    const int cls = comp ? 1 : 0;
    const uint64_t ngroups_total = a.first_group[a.nscans];
    unsigned long long *const desc_a = state + 2, *const desc = state + 2 + ngroups_total, *const tails = state + 2 + 2 * ngroups_total;
    unsigned long long *const sup_all = state + 2 + 3 * ngroups_total;
    unsigned long long *const host_abort = host_totals ? host_totals + 3 : nullptr;
    const bool last_group = (local + 1) * kGroup >= nblocks_chain;
    const int16_t *const comp_base = comp == 0 ? a.y : (comp == 1 ? a.cb : a.cr);
    // ---- the block, the tables, the DC predictors (jpeg/mod.rs:1268-1300: the first coefficient of the block before); the walk
    // of the AC bands (progressive.rs:141-210) into the lane's scratch from bit 0.  What the lane knows about itself (its index, its
    // block's address, whether it has one) is derived HERE for the walk and AGAIN behind it: the walk runs at the kernel's limit of
    // 80 vector registers, and every value alive across it was spilled to scratch memory (eight dwords a lane: 19 MB of writes a launch).
    uint32_t first_bits, all_bits;
    bool any[2], ends_zero[2];
    {
        const int lane = threadIdx.x;
        const uint64_t my_first = local * kGroup + (uint64_t)lane;
        const int16_t *const my_block = comp_base + (my_first < nblocks_chain ? my_first : 0) * 64;
        uint32_t w[32];
        const v4u *p4 = reinterpret_cast<const v4u *>(my_block);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const v4u q = p4[r];
            w[4 * r] = q.x; w[4 * r + 1] = q.y; w[4 * r + 2] = q.z; w[4 * r + 3] = q.w;
        }
        if (lane == 0) s_ext_dc = my_first ? *(my_block - 64) : (int16_t)0;
        for (int i = lane; i < kWalkClassWords; i += kGroup) tab[i] = a.tables[kTableWords + cls * kWalkClassWords + i];
        if (lane < 15) eobs[lane] = a.tables[cls * kClassSyms + kDcSyms + (lane << 4)];
        for (uint64_t i = (uint64_t)blockIdx.x * kGroup + lane; i < clear_words; i += (uint64_t)gridDim.x * kGroup) clear[i] = 0;
        s_dc[lane] = (int16_t)coef_of(w, 0);
        __syncthreads();
        PIXO_STAMP(1);
        FlatPack<LaneSink> p;
        p.sink = LaneSink{scratch + lane * kScratchPitch};
        p.acc = 0; p.pending = 0; p.word = 0;
        bands_pack_flat(w, split, tab, p, &first_bits, any, ends_zero);
        all_bits = p.word * 32u + p.pending;
        p.finish();
    }
    PIXO_STAMP(2);
    int lane_behind = threadIdx.x;
    asm volatile("" : "+v"(lane_behind));
    const int lane = lane_behind, wave = lane >> 6;
    const uint64_t my_first = local * kGroup + (uint64_t)lane;
    const bool live = my_first < nblocks_chain;
    const int16_t *const my_block = comp_base + (live ? my_first : 0) * 64;
    // the DC symbol (encode_dc_first, progressive.rs:112-133, al = 0) as bits of a word
    uint32_t dc_left, dc_len;
    {
        const DcBits db = dc_symbol_bits((int)s_dc[lane], lane ? (int)s_dc[lane - 1] : (int)s_ext_dc, tab);
        dc_left = db.left; dc_len = db.len;
    }
    if (!live) { dc_len = 0; first_bits = all_bits = 0; any[0] = any[1] = ends_zero[0] = ends_zero[1] = false; }
    const bool long_block = all_bits > kScratchWords * 32u;
    // what scan S of the component (0: DC, 1: the first AC band, 2: the second) is in the launch
    auto scan_k = [&](int S) -> uint32_t { return a.comp_pass[comp][S]; };
    // ---- the run counters of the AC scans: what each puts in front of and behind the lane's symbols.  Band S + 1's ballots, the
    // wavefronts' summaries, the group's count for the groups behind; its look-back runs on wavefront S
    BandEdge edge[2];
    {
        BandCount bc[2];
#pragma unroll
        for (int S = 0; S < 2; S++) {
            edge[S].pre.left = edge[S].pre.len = edge[S].post.left = edge[S].post.len = 0;
            if (S == 1 && !split) continue; // (uniform)
            const uint64_t zmask = PIXO_BALLOT64(any[S]), tmask = PIXO_BALLOT64(ends_zero[S]);
            bc[S] = band_count_in_wave(zmask, tmask, lane & 63);
            if ((lane & 63) == 0) {
                bool wany; uint32_t tail;
                band_wave_summary(zmask, tmask, &wany, &tail);
                wave_any[S][wave] = wany ? 1u : 0u;
                wave_tail[S][wave] = tail;
            }
        }
        __syncthreads();
        uint32_t into_wave[2] = {0, 0};
        bool wave_open[2] = {true, true};
#pragma unroll
        for (int S = 0; S < 2; S++) {
            if (S == 1 && !split) continue;
            uint32_t group_tail = 0; // the count behind the group's last non-empty block (all of its t when it holds none), not counting what flows in
            bool group_open = true;  // "no non-empty block so far": what flows into the group is still to be added
#pragma unroll
            for (int i = 0; i < kGroupWaves; i++) {
                if (i < wave) { if (wave_any[S][i]) { into_wave[S] = wave_tail[S][i]; wave_open[S] = false; } else into_wave[S] += wave_tail[S][i]; }
                if (wave_any[S][i]) { group_tail = wave_tail[S][i]; group_open = false; } else group_tail += wave_tail[S][i];
            }
            if (wave == S) { // (uniform per wavefront)
                const uint64_t floor_g = a.first_group[scan_k(S + 1)], g = floor_g + local;
                if ((lane & 63) == 0) {
                    if (!group_open) store_relaxed(&desc_a[g], kFlagPrefix | group_tail); // nothing before this group matters behind it
                    else publish_aggregate(desc_a, g, floor_g, group_tail);
                }
                const uint64_t sum = look_back(desc_a, g, floor_g, group_tail, state, host_abort, spin_budget);
                if ((lane & 63) == 0) {
                    if (sum == kLookBackFailed) s_abort = 1;
                    s_in[S] = sum;
                    if (group_open && g != floor_g && sum != kLookBackFailed) store_relaxed(&desc_a[g], kFlagPrefix | (sum + group_tail));
                }
            }
        }
        __syncthreads();
        PIXO_STAMP(3);
        if (s_abort) return;
#pragma unroll
        for (int S = 0; S < 2; S++) {
            if (S == 1 && !split) continue;
            const uint32_t C = bc[S].local + (bc[S].carried ? into_wave[S] + (wave_open[S] ? (uint32_t)s_in[S] : 0u) : 0u);
            const bool last_of_scan = my_first + 1 == nblocks_chain;
            edge[S] = band_edge(C, live, any[S], ends_zero[S], last_of_scan, eobs);
        }
    }
    // ---- every scan's bits of the lane, their exclusive prefixes in the group, the group's counts
    const uint32_t own[3] = {dc_len, first_bits, all_bits - first_bits};
    const uint32_t len[3] = {dc_len, edge[0].pre.len + own[1] + edge[0].post.len, split ? edge[1].pre.len + own[2] + edge[1].post.len : 0u};
    uint32_t my_bit[3], group_bits[3];
    uint32_t group_long = 0;
    {
        // (the DC scan's lengths — at most 27 bits a lane — ride in the top twelve bits of the first band's scan)
        const uint32_t incl01 = wave_inclusive_scan((len[0] << 20) | len[1]);
        const uint32_t incl2 = split ? wave_inclusive_scan(len[2]) : 0u; // (uniform)
        if ((lane & 63) == 63) { wave_sum[0][wave] = incl01 >> 20; wave_sum[1][wave] = incl01 & 0xFFFFFu; wave_sum[2][wave] = incl2; }
        const bool any_long = PIXO_ANY64(long_block);
        if ((lane & 63) == 0) wave_long[wave] = any_long ? 1u : 0u;
        __syncthreads();
        const uint32_t incl[3] = {incl01 >> 20, incl01 & 0xFFFFFu, incl2};
#pragma unroll
        for (int S = 0; S < 3; S++) {
            uint32_t wave_base = 0;
            group_bits[S] = 0;
#pragma unroll
            for (int i = 0; i < kGroupWaves; i++) {
                if (i < wave) wave_base += wave_sum[S][i];
                group_bits[S] += wave_sum[S][i];
            }
            my_bit[S] = wave_base + (incl[S] - len[S]);
        }
#pragma unroll
        for (int i = 0; i < kGroupWaves; i++) group_long |= wave_long[i];
    }
    // ---- where each scan's bits of this group begin: the aggregate out, the look-back — scan S on wavefront S, side by side
#pragma unroll
    for (int S = 0; S < 3; S++) {
        if (wave != S || (uint32_t)S >= nseg) continue; // (uniform per wavefront)
        const uint32_t k = scan_k(S);
        const uint64_t floor_g = a.first_group[k], g = floor_g + local;
        if ((lane & 63) == 0) publish_aggregate(desc, g, floor_g, group_bits[S]);
        const uint64_t sum = look_back_blocks(desc, sup_all + (floor_g >> 6) + k, g, floor_g, group_bits[S], state, host_abort, spin_budget);
        if ((lane & 63) == 0) {
            if (sum == kLookBackFailed) s_abort = 1;
            s_before[S] = sum;
            if (last_group && sum != kLookBackFailed) seg.bits[k] = sum + group_bits[S];
        }
    }
    __syncthreads();
    PIXO_STAMP(4);
    if (s_abort) return;
    // ---- place and write out, scan after scan through the same window.  What is left for the end: the word a scan's group
    // shares with the bits before it (s_head[S], s_first_word[S]; s_share[S]: 0 none, 1 hand the bits on, 2 complete the word)
    auto place = [&](auto tag) __attribute__((always_inline)) {
        constexpr int S = decltype(tag)::value;
        const uint32_t k = scan_k(S);
        const uint64_t floor_g = a.first_group[k], g = floor_g + local;
        uint32_t *const stream = stream0 + seg.var_word[k];
        const uint32_t gbits = group_bits[S];
        // The stream word two groups share travels as the earlier group's `tail`.  A group WITHOUT bits (192 empty blocks) is
        // transparent for it: it says so at once, and the group that needs the word finds the last group with bits by a
        // look-back over the tails (64 per round, kLookBatch = 1) — passing the word on from group to group made one chain of waits out
        // of every run of empty groups (a smooth 4096x4096 image: 1,366 groups in a row, 276 us for the launch).
        if (gbits == 0 && !last_group) { // nothing to place: its B descriptor stays an aggregate of zero bits, which later groups walk past
            if (lane == 0) store_relaxed(&tails[g], kFlagAggregate);
            return;
        }
        const BandBits none{0u, 0u};
        const BandBits pre = S ? edge[S ? S - 1 : 0].pre : none, post = S ? edge[S ? S - 1 : 0].post : none;
        const uint32_t from = S == 2 ? first_bits : 0u; // where the scan's own bits begin in the lane's scratch
        const uint32_t bit_words = (gbits + 31) >> 5;
        const uint32_t local_words = bit_words ? bit_words : 1u; // (a group without bits still runs one round: shared word)
        const uint64_t start = s_before[S];
        uint64_t end = start + gbits;
        uint32_t pad_word = ~0u, pad_mask = 0;
        if (last_group) { // every scan ends on a byte boundary, padded with 1-bits (BitWriterMsb::finish)
            const uint32_t n = (uint32_t)((8 - (end & 7)) & 7);
            if (n) {
                pad_word = (uint32_t)((end >> 5) - (start >> 5));
                pad_mask = ((1u << n) - 1u) << (32 - (uint32_t)(end & 31) - n);
            }
            end += n;
        }
        const uint64_t first_word = start >> 5;
        const uint32_t sh = (uint32_t)(start & 31);
        const uint32_t out_words = (uint32_t)((end - (first_word << 5) + 31) >> 5);
        const bool tail_partial = (end & 31) != 0 && !last_group;
        // (every group leaves SOMETHING in its tail slot — the look-back over the tails waits for all the slots it reads:
        // a group whose bits end on a word boundary hands nothing on)
        if (!tail_partial && lane == 0) store_relaxed(&tails[g], kFlagPrefix);
        if (lane == 0) s_carry = 0;
        uint32_t head_word = 0;
        for (uint32_t wbase = 0; wbase < local_words; wbase += kWindowWords) {
            const uint32_t wn = local_words - wbase < kWindowWords ? local_words - wbase : kWindowWords;
            for (uint32_t i = lane; i < wn; i += kGroup) buf[i] = 0;
            __syncthreads();
            const uint32_t dummy = kWindowWords + (uint32_t)lane;
            // `n` bits at the top of `left`, at bit `at` of the window (negative / beyond it: the parts outside go to the dummy word)
            auto or_bits = [&](uint32_t left, uint32_t n, int64_t at) {
                if (!PIXO_ANY64(n != 0)) return; // (wave-uniform)
                const uint32_t bsh = (uint32_t)(at & 31), d = (uint32_t)(at >> 5);
                (void)__hip_atomic_fetch_or(&buf[(n && d < wn) ? d : dummy], left >> bsh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_or(&buf[(n && d + 1 < wn) ? d + 1 : dummy], bsh ? left << (32 - bsh) : 0u, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            const int64_t rel = (int64_t)my_bit[S] - (int64_t)wbase * 32; // the lane's first bit
            const int64_t rel_own = rel + (int64_t)pre.len;                // its own symbols' first bit
            if (S == 0) {
                or_bits(dc_left, dc_len, rel);
            } else {
                or_bits(pre.left, pre.len, rel);
                or_bits(post.left, post.len, rel_own + (int64_t)own[S]);
                if (!group_long) { // the scan's part of the lane's scratch, word by word (two LDS ORs each)
                    // (the length as a value of this round: the twelve words' masks are not computed in front of the rounds' loop and
                    // kept — eight of them in scratch memory — but where they are used)
                    uint32_t n_own = own[S];
                    asm volatile("" : "+v"(n_own));
                    const uint32_t nw = (n_own + 31) >> 5, bsh = (uint32_t)(rel_own & 31);
                    const uint32_t d0 = (uint32_t)(rel_own >> 5);
#pragma unroll
                    for (uint32_t j = 0; j < kScratchWords; j++) {
                        if (!PIXO_ANY64(j < nw)) break; // (wave-uniform)
                        const uint32_t v = scratch_bits_word(scratch + lane * kScratchPitch, kScratchWords, from, n_own, j);
                        const uint32_t d = d0 + j;
                        (void)__hip_atomic_fetch_or(&buf[d < wn ? d : dummy], v >> bsh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        (void)__hip_atomic_fetch_or(&buf[d + 1 < wn ? d + 1 : dummy], bsh ? v << (32 - bsh) : 0u, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else if (PIXO_ANY64(live && rel_own < (int64_t)wn * 32 && rel_own + (int64_t)own[S] > 0)) {
                    // a group with a block too long for the scratch (rare: noise at q >= 90): its scans' bits come from a second
                    // walk of the band straight into the window — the block is read again (L2)
                    uint32_t w[32];
                    const v4u *p4 = reinterpret_cast<const v4u *>(my_block);
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        const v4u q = p4[r];
                        w[4 * r] = q.x; w[4 * r + 1] = q.y; w[4 * r + 2] = q.z; w[4 * r + 3] = q.w;
                    }
                    FlatPack<LdsSink> p;
                    p.sink = LdsSink{buf, live ? wn : 0u, dummy};
                    p.acc = 0;
                    p.pending = (uint32_t)(rel_own & 31);
                    p.word = (uint32_t)(rel_own >> 5);
                    bool z2, t2;
                    band_pack_flat(w, S == 2 ? 11 : 1, (S == 1 && split) ? 10 : 63, tab, p, &z2, &t2);
                    p.finish();
                }
            }
            __syncthreads();
            const uint32_t carry = s_carry;
            const bool last_round = wbase + wn == local_words;
            const uint32_t upto = last_round ? out_words - wbase : wn;
            uint32_t word0 = 0, tail_word = 0;
            for (uint32_t i = lane; i < upto; i += kGroup) {
                const uint32_t j = wbase + i;
                const uint32_t cur = i < wn ? buf[i] : 0u, prev = i ? buf[i - 1] : carry;
                uint32_t v = sh ? (cur >> sh) | (prev << (32 - sh)) : cur;
                v |= j == pad_word ? pad_mask : 0u;
                const bool is_head = j == 0 && sh != 0, is_tail = tail_partial && j + 1 == out_words;
                if (is_head) word0 = v;
                if (is_tail) tail_word = v;
                if (!is_head && !is_tail) __builtin_nontemporal_store(v, &stream[first_word + j]);
            }
            const bool has_tail = last_round && tail_partial;
            const bool pass_through = has_tail && out_words == 1 && sh != 0;
            if (has_tail && !pass_through && (uint32_t)lane == (upto - 1) % kGroup) store_relaxed(&tails[g], kFlagPrefix | tail_word);
            if (wbase == 0) head_word = word0;
            if (lane == 0) s_carry = buf[wn - 1];
            __syncthreads();
        }
        // ---- the word shared with the bits before.  A group whose bits all lie INSIDE that word (a few symbols among empty
        // blocks) has nothing to write: it leaves its bits as an AGGREGATE in its tail slot — bits of different groups are
        // different bits, so the look-back's sum over such slots is their OR — and the group that completes the word finds
        // them, and the tail of the last group that left a word unfinished (a PREFIX), in one look-back: no group waits for
        // another group's look-back (handing the word on from group to group chained every run of small groups).
        if (sh != 0 && lane == 0) { // (lane 0 holds word 0 of the first round)
            s_head[S] = head_word;
            s_first_word[S] = first_word;
            s_share[S] = (tail_partial && out_words == 1) ? 1u : 2u;
        }
    };
    place(std::integral_constant<int, 0>{});
    PIXO_STAMP(5);
    place(std::integral_constant<int, 1>{});
    PIXO_STAMP(6);
    if (split) place(std::integral_constant<int, 2>{});
    // ---- the shared words: scan S on wavefront S, side by side
    __syncthreads();
#pragma unroll
    for (int S = 0; S < 3; S++) {
        if (wave != S || (uint32_t)S >= nseg) continue;
        const uint32_t k = scan_k(S);
        const uint64_t floor_g = a.first_group[k], g = floor_g + local;
        const uint32_t hw = s_head[S];
        const uint32_t how = s_share[S];
        if (how == 1u) {
            if ((lane & 63) == 0) store_relaxed(&tails[g], kFlagAggregate | hw);
        } else if (how == 2u) {
            const uint64_t inherited = look_back(tails, g, floor_g, 0, state, host_abort, spin_budget);
            if (inherited != kLookBackFailed && (lane & 63) == 0)
                __builtin_nontemporal_store((uint32_t)inherited | hw, &(stream0 + seg.var_word[k])[s_first_word[S]]);
        }
    }
    PIXO_STAMP(7);
}

// ---- count: symbol statistics for optimised tables (count_block, jpeg/mod.rs:826-860) with the flat walk ---------------
// One lane per block like the code kernel; the counters live in LDS per workgroup (LDS atomics without return, a
// private dummy counter per lane for "no symbol here").  Every workgroup then stores its counters as one row of a
// slab in HBM and a second, tiny kernel adds the rows up: 2048 workgroups adding their 536 counters to the same 536
// words with global atomics took longer than the walk itself (58 us for an image with hardly any symbols).
// Restart intervals reset the predictors exactly as in the scan (jpeg/mod.rs:1441-1444).
constexpr unsigned kCountWorkgroups = 2048; // rows of the slab (a larger scan: several groups per workgroup)
constexpr int kCountRowsPerSum = 64; // rows one workgroup of the summing kernel adds up (16 per thread)
struct LdsBump {
    uint32_t *hist; // the class's kWalkClassWords counters
    uint32_t dummy; // index of this lane's dummy counter, relative to hist
    bool live;
    __device__ __forceinline__ void bump(uint32_t slot, bool on, uint32_t amount)
    {
        (void)__hip_atomic_fetch_add(&hist[on && live ? slot : dummy], amount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

template <int MODE>
__global__ __launch_bounds__(kGroup) void scan_count_kernel(const ScanArgs a, uint32_t *slab, unsigned long long *hist)
{
    __shared__ uint32_t lhist[kWalkWords + kGroup];
    const int lane = threadIdx.x;
    for (int i = lane; i < kWalkWords + kGroup; i += kGroup) lhist[i] = 0;
    if (blockIdx.x == 0) // (the sums start from zero: cheaper here than a memset launch)
        for (int i = lane; i < kTableWords; i += kGroup) hist[i] = 0;
    __syncthreads();
    const uint64_t ngroups = (a.nblocks + kGroup - 1) / kGroup;
    for (uint64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const uint64_t s = g * kGroup + lane;
        const bool live = s < a.nblocks;
        const BlockRef ref = block_of(MODE, live ? s : 0);
        const int16_t *base = ref.comp == 0 ? a.y : (ref.comp == 1 ? a.cb : a.cr);
        uint32_t w[32];
        {
            const v4u *p = reinterpret_cast<const v4u *>(base + ref.index * 64);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const v4u q = p[r];
                w[4 * r] = q.x; w[4 * r + 1] = q.y; w[4 * r + 2] = q.z; w[4 * r + 3] = q.w;
            }
        }
        int prev_dc = ref.index ? (int)base[(ref.index - 1) * 64] : (int)a.seed_dc[ref.comp];
        if (a.restart) {
            const uint64_t mcu = s / a.blocks_per_mcu;
            const uint32_t k = (uint32_t)(s - mcu * a.blocks_per_mcu);
            const bool first_of_comp = MODE == 2 ? (k == 0 || k >= 4) : true;
            if (mcu % a.restart == 0 && first_of_comp) prev_dc = 0;
        }
        const uint32_t cls = ref.comp == 0 ? 0u : 1u;
        LdsBump h{lhist + cls * kWalkClassWords, (uint32_t)(kWalkWords + lane) - cls * kWalkClassWords, live};
        block_count_flat(w, prev_dc, h);
    }
    __syncthreads();
    for (int i = lane; i < kWalkWords; i += kGroup) slab[(size_t)blockIdx.x * kWalkWords + i] = lhist[i];
}

// blockIdx.x: 64 columns of the slab, blockIdx.y: kCountRowsPerSum rows; thread = column x one of four row phases
__global__ __launch_bounds__(256) void scan_count_sum_kernel(const uint32_t *slab, uint32_t rows, unsigned long long *hist)
{
    __shared__ uint32_t part[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), phase = threadIdx.x >> 6;
    const uint32_t r0 = blockIdx.y * kCountRowsPerSum;
    uint32_t n = 0;
    if (col < kWalkWords) {
#pragma unroll
        for (int k = 0; k < kCountRowsPerSum / 4; k++) {
            const uint32_t r = r0 + 4 * k + phase;
            n += r < rows ? slab[(size_t)r * kWalkWords + col] : 0u;
        }
    }
    part[phase][threadIdx.x & 63] = n;
    __syncthreads();
    if (phase == 0 && col < kWalkWords) {
        const unsigned long long sum = (unsigned long long)part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        const int sym = walk_slot_symbol(col % kWalkClassWords);
        if (sum && sym >= 0) atomicAdd(&hist[(col / kWalkClassWords) * kClassSyms + sym], sum);
    }
}

// ---- stuff: 16 KiB tiles of the packed stream ----------------------------------------------------------------------
#ifndef PIXO_STUFF_LANE_WORDS
#define PIXO_STUFF_LANE_WORDS 16 // (A/B builds: tools/ab_build.sh; 16 words per lane = 16 KiB tiles)
#endif
constexpr int kStuffThreads = 256, kLaneWords = PIXO_STUFF_LANE_WORDS, kWaveBytes = 64 * kLaneWords * 4, kTileBytes = (kStuffThreads / 64) * kWaveBytes;
constexpr uint32_t kMaxSegGap = 1024;                        // bytes a segmented scan may leave free behind a segment (SegArgs::marker_bytes)
constexpr uint32_t kStageBytes = 2 * kTileBytes + 32 + kMaxSegGap; // worst case: every byte 0xFF, + the output's alignment skew, + the gap behind a segment

// ---- segmented scans: where every segment's tiles begin -----------------------------------------------------------------
// One workgroup, after scan_code<SEG>: from the segments' bit lengths the bytes of each (1-padded: whole bytes) and the
// exclusive prefix sum of their 16 KiB tile counts, so that a stuffing workgroup finds its segment by binary search.
// layout: [0] total tiles, [1 ..] first tile of segment i (nsegs + 1 entries); bytes: packed bytes of segment i.
__global__ __launch_bounds__(1024) void seg_layout_kernel(const unsigned long long *seg_bits, uint64_t nsegs, unsigned long long *layout,
                                                        unsigned long long *bytes, unsigned long long *host_totals)
{
    __shared__ uint64_t carry;
    __shared__ uint64_t wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < nsegs; base += 1024) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t nb = i < nsegs ? (seg_bits[i] + 7) / 8 : 0;
        if (i < nsegs) bytes[i] = nb;
        const uint32_t t = (uint32_t)((nb + kTileBytes - 1) / kTileBytes);
        const uint32_t incl = wave_inclusive_scan(t);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint64_t before = carry;
        for (int k = 0; k < wave; k++) before += wsum[k];
        if (i < nsegs) layout[1 + i] = before + incl - t;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        layout[1 + nsegs] = carry;
        layout[0] = carry;
        if (host_totals) host_totals[2] = carry; // (the host launches the tiles its guess missed)
    }
}

// SEG: the tiles of all segments in one grid.  A tile's aggregate is what it PRODUCES (its bytes + the zeros stuffed
// into them + the two marker bytes behind a segment's last tile), so the look-back over all tiles before it gives its
// place in the output whatever the segments' lengths; the last tile of segment k also stores where the segment ends
// (seg.out_end[k], device and pinned mailbox: a batch's files are cut there) and — restart intervals — writes RSTn.
template <bool SEG>
__global__ __launch_bounds__(kStuffThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void stuff_fused_kernel(const uint32_t *stream, const unsigned long long *code_state,
                                                                   uint32_t shift, uint32_t band, unsigned long long *state,
                                                                   uint8_t *out, uint32_t out_skew, uint64_t out_cap, uint64_t tile_offset,
                                                                   unsigned long long *clear, uint32_t clear_words,
                                                                   unsigned long long *host_totals, unsigned long long *out_chain,
                                                                   uint32_t piece, const SegArgs seg, uint32_t spin_budget, const GateMark gate)
{
    if (threadIdx.x == 0) dispatch_mark(gate.slots, gate.seq, blockIdx.x, gridDim.x);
    // The bytes to stuff are the stream's bits from bit `shift` (< 8) on: 0 for a whole image (all bytes, the padded
    // last one included); for a band that starts inside a byte of the scan, its first `shift` bits belong to the byte
    // it shares with the band before, its whole bytes follow, the bits left over are the next band's business.
    // state: [0] abort flag, [1] stuffed bytes (out), [2] bytes of the packed stream that were stuffed (out), [3 ..] descriptors
    // out: 16-byte aligned; the first stuffed byte goes to out[out_skew] (< 16; the bytes before it are somebody else's — the
    // file headers when `out` is the caller's host buffer — and are not touched); out_cap counts from out[0].
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStageBytes + kStuffThreads];
    __shared__ uint32_t wave_sum[kStuffThreads / 64];
    __shared__ unsigned long long s_before;
    __shared__ uint32_t s_abort;
    const int lane = threadIdx.x, wave = lane >> 6;
    if (lane == 0) s_abort = 0;
    // (housekeeping for the NEXT scan_code launch: its descriptors — everything but word 1, the length — back to zero)
    for (uint64_t i = (uint64_t)blockIdx.x * kStuffThreads + lane; i < clear_words; i += (uint64_t)gridDim.x * kStuffThreads)
        if (i != 1) clear[i] = 0;
    // (a scan stuffed piece by piece: this piece's bytes go behind those of the pieces before, out_chain[piece])
    const uint64_t out_before = out_chain && piece ? out_chain[piece] : 0ull;
    const uint64_t t = tile_offset + blockIdx.x; // one tile per workgroup (the launcher guesses how many there are)
    uint64_t nbytes, ntiles, local_t = t, sidx = 0;
    if (SEG && seg.var) {
        // the scans of a progressive file (at most eight segments of different sizes): every workgroup works the layout out itself
        // from the segments' bit counts — round 4 ran a one-workgroup kernel (seg_layout) between the coder and this one for it:
        // 5 us and a launch per file
        uint64_t first = 0;
        bool found = false;
        nbytes = 0;
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            if (i >= seg.nsegs) continue; // (uniform)
            const uint64_t nb = (seg.bits[i] + 7) / 8, tl = (nb + kTileBytes - 1) / kTileBytes;
            if (!found && t < first + tl) { sidx = i; local_t = t - first; nbytes = nb; found = true; }
            first += tl;
        }
        ntiles = first;
        if (blockIdx.x == 0 && lane == 0 && tile_offset == 0 && host_totals) host_totals[2] = ntiles; // (the host launches the tiles its guess missed)
        if (!found) return;
        stream += seg.var_word[sidx];
    } else if (SEG) {
        ntiles = seg.layout[0];
        if (t >= ntiles) return;
        uint64_t lo = 0, hi = seg.nsegs; // the segment whose tiles include t: first[lo] <= t < first[lo + 1]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) / 2;
            if (seg.layout[1 + mid] <= t) lo = mid; else hi = mid;
        }
        sidx = lo;
        local_t = t - seg.layout[1 + sidx];
        nbytes = seg.bytes[sidx];
        stream += seg.var ? seg.var_word[sidx] : sidx * seg.stream_words;
    } else {
        const uint64_t total_bits = code_state[1];
        nbytes = band ? (total_bits - (shift < total_bits ? shift : total_bits)) / 8 : (total_bits + 7) / 8;
        ntiles = (nbytes + kTileBytes - 1) / kTileBytes;
    }
    unsigned long long *desc = state + 3;
    if (ntiles == 0) {
        if (blockIdx.x == 0 && lane == 0 && tile_offset == 0) {
            state[1] = 0; state[2] = 0;
            if (host_totals) { host_totals[1] = 0; host_totals[2] = 0; }
            if (out_chain) out_chain[piece + 1] = out_before;
        }
        return;
    }
    if (t < ntiles) {
        for (uint32_t i = 16u * lane; i < kStageBytes; i += 16u * kStuffThreads) *reinterpret_cast<v4u *>(stage + i) = v4u{0, 0, 0, 0};
        // ---- in.  Word k of lane l (of wavefront v) is word 64 k + l of the wavefront's 4 KiB: every load instruction
        // reads 256 consecutive bytes, and — what matters — the byte stores into the stage below hit 64 different banks.
        // (Round 2's first form gave every lane 64 CONSECUTIVE bytes: lanes 16 words apart, four banks for a wavefront,
        // every one of the 64 byte stores a 16-way conflict — the whole 20 us of the kernel.)
        const int wl = lane & 63;
        const uint64_t wave0 = local_t * kTileBytes + (uint64_t)wave * kWaveBytes; // first byte of this wavefront's part
        uint32_t w[kLaneWords];
        uint64_t flags = 0; // four flags per word, stream order (bit 4 k + b = byte b of word k is 0xFF and exists)
        {
            const uint32_t *p = stream + wave0 / 4 + wl; // (the stream buffer has 64 bytes of slack behind its last word)
#pragma unroll
            for (int k = 0; k < kLaneWords; k++) {
                const uint64_t at = wave0 + 256u * k + 4u * wl;
                uint32_t x = 0;
                if (at < nbytes) {
                    x = p[64 * k];
                    // funnel: byte i of the output is bits [8 i + shift, 8 i + shift + 8) of the stream
                    if (shift) x = (x << shift) | (p[64 * k + 1] >> (32 - shift));
                }
                w[k] = x;
                const uint32_t exist = at < nbytes ? (nbytes - at < 4 ? (uint32_t)(nbytes - at) : 4u) : 0u;
                const uint32_t m = (zero_byte_mask(~x) >> 7) & 0x01010101u; // bits 24, 16, 8, 0 = stream bytes 0, 1, 2, 3 of the word
                const uint32_t m4 = ((m * 0x08040201u) >> 24) & ((1u << exist) - 1u);
                flags |= (uint64_t)m4 << (4 * k);
            }
        }
        // 0xFF bytes before every word, in stream order = row by row (a row: word k of the 64 lanes): the counts of two
        // rows share one 32-bit scan (a row holds at most 256), the rows' totals chain through scalar registers
        uint32_t before[kLaneWords]; // 0xFF bytes of this wavefront before word k of this lane
        uint32_t wave_ff = 0;
#pragma unroll
        for (int k = 0; k < kLaneWords; k += 2) {
            const uint32_t c0 = (uint32_t)__builtin_popcount((uint32_t)(flags >> (4 * k)) & 0xFu);
            const uint32_t c1 = (uint32_t)__builtin_popcount((uint32_t)(flags >> (4 * k + 4)) & 0xFu);
            const uint32_t sc = wave_inclusive_scan(c0 | (c1 << 16));
            const uint32_t rows = (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
            before[k] = wave_ff + (sc & 0xFFFFu) - c0;
            wave_ff += rows & 0xFFFFu;
            before[k + 1] = wave_ff + (sc >> 16) - c1;
            wave_ff += rows >> 16;
        }
        if (wl == 0) wave_sum[wave] = wave_ff;
        __syncthreads();
        uint32_t wave_base = 0, tile_ff = 0;
#pragma unroll
        for (int k = 0; k < kStuffThreads / 64; k++) {
            if (k < wave) wave_base += wave_sum[k];
            tile_ff += wave_sum[k];
        }
        const uint64_t tile_in = nbytes - local_t * kTileBytes < (uint64_t)kTileBytes ? nbytes - local_t * kTileBytes : (uint64_t)kTileBytes;
        // SEG: is this the segment's last tile, and does a marker follow the segment?
        const bool seg_last = SEG && (local_t + 1) * kTileBytes >= nbytes;
        // (restart intervals: the two bytes of RSTn; a batch: room for the next file's headers — the caller's, written by the
        // host — so that the whole batch is ONE run of bytes with every file at its final distance from the others)
        const uint32_t marker = seg_last && sidx + 1 < seg.nsegs ? seg.marker_bytes : 0u;
        // the tile's aggregate: 0xFF bytes (one scan: its input bytes are known from t) — SEG: everything it produces
        const uint32_t aggregate = SEG ? (uint32_t)tile_in + tile_ff + marker : tile_ff;
        if (lane == 0) publish_aggregate(desc, t, 0, aggregate);
        if (wave == 0) {
            const uint64_t sum = look_back(desc, t, 0, aggregate, state, host_totals ? host_totals + 3 : nullptr, spin_budget);
            if (lane == 0) {
                if (sum == kLookBackFailed) s_abort = 1;
                s_before = sum;
                if (t) store_relaxed(&desc[t], kFlagPrefix | (sum + aggregate));
            }
        }
        __syncthreads();
        if (s_abort) return; // (the look-back gave up: the host codes this scan again, see Waiter)
        const uint64_t produced_before = SEG ? s_before : t * kTileBytes + s_before;
        const uint64_t dst0 = out_skew + out_before + produced_before; // where the tile's first output byte goes
        const uint32_t tile_out = (uint32_t)tile_in + tile_ff + marker; // bytes the tile produces
        if (SEG && seg_last && lane == 0) { // where the segment's entropy-coded bytes end (its marker not counted)
            const uint64_t end = dst0 + tile_out - marker - out_skew;
            seg.out_end[sidx] = end;
            if (seg.host_out_end) seg.host_out_end[sidx] = end;
        }
        if (t + 1 == ntiles && lane == 0) { // (totals of this launch's piece; out_chain: of the scan so far)
            state[1] = dst0 + tile_out - out_skew - out_before; state[2] = nbytes;
            if (host_totals) { host_totals[1] = dst0 + tile_out - out_skew - out_before; if (!SEG) host_totals[2] = nbytes; }
            if (out_chain) out_chain[piece + 1] = dst0 + tile_out - out_skew;
        }
        // expand into LDS at the output's alignment (LDS dwords = global dwords).  The stage was zeroed: only the
        // stream's bytes are written, each moved up by the number of 0xFF bytes before it — the gaps ARE the stuffed
        // zeros.  No branch per byte: bytes that do not exist go to a dummy byte.
        const uint32_t skew = (uint32_t)(dst0 & 15);
        const uint32_t at0 = skew + (uint32_t)wave * kWaveBytes + wave_base + 4u * wl;
#pragma unroll
        for (int k = 0; k < kLaneWords; k++) {
            const uint64_t at = wave0 + 256u * k + 4u * wl;
            const uint32_t exist = at < nbytes ? (nbytes - at < 4 ? (uint32_t)(nbytes - at) : 4u) : 0u;
            const uint32_t m4 = (uint32_t)(flags >> (4 * k)) & 0xFu;
            const uint32_t to = at0 + 256u * k + before[k];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t byte = (w[k] >> (24 - 8 * b)) & 0xFFu;
                const uint32_t moved = (uint32_t)__builtin_popcount(m4 & ((1u << b) - 1u));
                stage[(uint32_t)b < exist ? to + b + moved : kStageBytes + (uint32_t)lane] = (uint8_t)byte;
            }
        }
        __syncthreads();
        if (marker == 2 && seg.rst_markers && lane == 0) { // RSTn behind the segment (jpeg/mod.rs:1431-1440): FF D0 + (index & 7), never stuffed
            stage[skew + tile_out - 2] = 0xFF;
            stage[skew + tile_out - 1] = (uint8_t)(0xD0 + (sidx & 7));
        }
        if (marker) __syncthreads();
        // out: leading bytes up to the first aligned 16 bytes, aligned 16-byte pieces, trailing bytes
        const uint64_t base = dst0 - skew; // multiple of 16
        const uint32_t end = skew + tile_out;
        const uint32_t first_q = skew ? 16u : 0u, last_q = end & ~15u;
        if (first_q <= last_q) {
            for (uint32_t i = first_q + 16u * lane; i < last_q; i += 16u * kStuffThreads)
                if (base + i + 16 <= out_cap) __builtin_nontemporal_store(*reinterpret_cast<const v4u *>(stage + i), reinterpret_cast<v4u *>(out + base + i));
            if (lane < 16) { // bytes [skew, min(16, end)) and [last_q, end)
                const uint32_t i = skew + lane;
                if (skew && i < 16 && i < end && base + i < out_cap) out[base + i] = stage[i];
                const uint32_t j = last_q + lane;
                if (j < end && j >= first_q && base + j < out_cap) out[base + j] = stage[j];
            }
        } else { // the whole tile lies inside one 16-byte piece
            if ((uint32_t)lane + skew < end && lane < 16 && base + skew + lane < out_cap) out[base + skew + lane] = stage[skew + lane];
        }
    }
}
} // namespace

size_t fused_code_state_words(uint64_t nblocks)
{ // abort flag, total bits, per group: descriptor + tail, per 64 groups: block sum (+ 1)
    const size_t groups = (size_t)((nblocks + kGroup - 1) / kGroup);
    const SupLayout sl = sup_layout(groups, (groups + 63) / 64 + 1);
    return 2 + 2 * groups + (size_t)sl.copies * sl.stride;
}
size_t fused_code_state_words_seg(uint64_t nsegs, uint64_t seg_blocks)
{
    const size_t per = (size_t)((seg_blocks + kGroup - 1) / kGroup);
    const SupLayout sl = sup_layout((uint64_t)nsegs * per, (uint64_t)nsegs * ((per + 63) / 64) + 1);
    return 2 + 2 * (size_t)nsegs * per + (size_t)sl.copies * sl.stride;
}
size_t fused_stuff_state_words(uint64_t max_stream_bytes) { return 3 + (size_t)((max_stream_bytes + kTileBytes - 1) / kTileBytes); }
uint32_t seg_groups(uint64_t seg_blocks) { return (uint32_t)((seg_blocks + kGroup - 1) / kGroup); }
uint32_t seg_max_gap() { return kMaxSegGap; }

hipError_t launch_scan_code(const ScanArgs &a, unsigned long long *d_state, bool state_is_zero, uint32_t *d_stream,
                            unsigned long long *d_clear, size_t clear_words, unsigned long long *host_totals, hipStream_t s,
                            const ScanPiece *piece_or_null, const SegArgs *seg_or_null, uint32_t spin_budget)
{
    const ScanPiece piece = piece_or_null ? *piece_or_null : ScanPiece{0, 0, nullptr, nullptr};
    const SegArgs seg = seg_or_null ? *seg_or_null : SegArgs{};
    const uint64_t ngroups = seg_or_null ? seg.nsegs * seg.groups : (a.nblocks + kGroup - 1) / kGroup;
    const size_t state_words = seg_or_null ? fused_code_state_words_seg(seg.nsegs, seg.blocks) : fused_code_state_words(a.nblocks);
    if (host_totals) host_totals[3] = 0; // (the kernels' abort flag: nothing in flight writes it, a context's launches are serial)
    if (ngroups == 0) {
        if (host_totals) host_totals[0] = 0;
        return hipMemsetAsync(d_state, 0, 16, s);
    }
    if (!state_is_zero) {
        hipError_t e = hipMemsetAsync(d_state, 0, state_words * 8, s);
        if (e != hipSuccess) return e;
    }
    if (ngroups > 0x7FFFFFFFull || clear_words > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)ngroups;
    const uint32_t cw = d_clear ? (uint32_t)clear_words : 0u;
    const DispatchGate gate(s, ngroups); // (held until the kernel is enqueued: dispatch_gate.hpp)
    const GateMark gm = gate.mark();
#define PIXO_LAUNCH_CODE(MODE, SEG) hipLaunchKernelGGL((scan_code_kernel<MODE, SEG>), dim3(grid), dim3(kGroup), 0, s, a, d_state, d_stream, d_clear, cw, host_totals, piece, seg, spin_budget, gm)
    if (seg_or_null) {
        if (a.mode == 2) PIXO_LAUNCH_CODE(2, true); else if (a.mode == 1) PIXO_LAUNCH_CODE(1, true); else PIXO_LAUNCH_CODE(0, true);
    } else {
        if (a.mode == 2) PIXO_LAUNCH_CODE(2, false); else if (a.mode == 1) PIXO_LAUNCH_CODE(1, false); else PIXO_LAUNCH_CODE(0, false);
    }
#undef PIXO_LAUNCH_CODE
    return hipGetLastError();
}

size_t prog_code_state_words(uint64_t groups) { return 2 + 3 * (size_t)groups + (size_t)(groups >> 6) + 8; } // (+ the B chains' block sums: one word per 64 groups and one per scan)
uint64_t prog_groups(uint32_t, uint64_t blocks) { return (blocks + kGroup - 1) / kGroup; }
size_t prog_stream_bytes(uint32_t scan_id, uint64_t blocks)
{ // per block: a DC symbol of at most 16 + 11 bits; a band of n coefficients: n symbols of at most 16 + 10 bits (ZRL codes
  // only where coefficients are missing) + run symbols of at most 30 bits in front and behind; + 64 bytes the kernels read into
    const uint64_t bits = scan_id < 3 ? 27 : ((scan_id == 3 ? 10 : (scan_id == 4 ? 53 : 63)) * 26 + 60);
    return (size_t)((blocks * bits + 7) / 8 + 64 + 15) / 16 * 16;
}
void prog_code_plan(ProgCode &a)
{
    for (int c = 0; c < 3; c++) { a.comp_blocks[c] = 0; a.comp_npass[c] = 0; }
    for (uint32_t k = 0; k < a.nscans; k++) {
        const int sc = (int)a.scan_id[k], c = sc < 3 ? sc : (sc <= 4 ? 0 : sc - 4); // (prog_comp)
        a.comp_blocks[c] = a.size[k];
        a.comp_pass[c][a.comp_npass[c]++] = k;
    }
    // the workgroups' order: the chrominance components first — their groups serve two scans where a luminance group serves
    // three, so they leave their slots early (luminance first measured the same within 1 %: profiles/r05_prog_code.txt)
    const uint32_t order[3] = {1, 2, 0};
    a.comp_first[0] = 0;
    for (int i = 0; i < 3; i++) {
        const uint32_t c = order[i];
        a.comp_order[i] = c;
        a.comp_first[i + 1] = a.comp_first[i] + (a.comp_npass[c] ? prog_groups(0, a.comp_blocks[c]) : 0);
    }
}
hipError_t launch_prog_code(const ProgCode &a, const SegArgs &seg, unsigned long long *d_state, bool state_is_zero, uint32_t *d_stream,
                            unsigned long long *d_clear, size_t clear_words, unsigned long long *host_totals, hipStream_t s, uint32_t spin_budget)
{
    const uint64_t groups = a.first_group[a.nscans], grid = a.comp_first[3];
    if (host_totals) host_totals[3] = 0;
    if (groups == 0 || grid == 0 || grid > 0x7FFFFFFFull || clear_words > 0xFFFFFFFFull) return hipErrorInvalidValue;
    if (!state_is_zero) {
        hipError_t e = hipMemsetAsync(d_state, 0, prog_code_state_words(groups) * 8, s);
        if (e != hipSuccess) return e;
    }
    const DispatchGate gate(s, grid); // (held until the kernel is enqueued: dispatch_gate.hpp)
    hipLaunchKernelGGL(prog_code_kernel, dim3((unsigned)grid), dim3(kGroup), 0, s, a, seg, d_state, d_stream, d_clear,
                       d_clear ? (uint32_t)clear_words : 0u, host_totals, spin_budget, gate.mark());
    return hipGetLastError();
}

// Files of a batch that stay in HBM (a caller that gathers them over RCCL): the scans already lie at their final spacing, what
// is missing are the bytes between them — EOI of file i - 1 and the headers of file i, which are the same for every file.
// `meta` = [offsets[0 .. batch - 1], end] as 64-bit words, then the header bytes.  One workgroup per seam.
__global__ __launch_bounds__(256) void batch_seams_kernel(uint8_t *arena, const unsigned long long *meta, uint32_t batch, uint32_t hdr)
{
    const uint32_t i = blockIdx.x; // seam i: in front of file i; seam `batch`: the last file's EOI
    const uint8_t *head = reinterpret_cast<const uint8_t *>(meta + batch + 1);
    const unsigned long long at = meta[i];
    if (i < batch)
        for (uint32_t k = threadIdx.x; k < hdr; k += 256) arena[at + k] = head[k];
    if (i > 0 && threadIdx.x < 2) arena[at - 2 + threadIdx.x] = threadIdx.x ? 0xD9 : 0xFF;
}

hipError_t launch_batch_seams(uint8_t *d_arena, const unsigned long long *d_meta, uint32_t batch, uint32_t hdr, hipStream_t s)
{
    hipLaunchKernelGGL(batch_seams_kernel, dim3(batch + 1), dim3(256), 0, s, d_arena, d_meta, batch, hdr);
    return hipGetLastError();
}

hipError_t launch_seg_layout(const SegArgs &seg, unsigned long long *d_layout, unsigned long long *d_bytes, unsigned long long *host_totals, hipStream_t s)
{
    hipLaunchKernelGGL(seg_layout_kernel, dim3(1), dim3(1024), 0, s, seg.bits, seg.nsegs, d_layout, d_bytes, host_totals);
    return hipGetLastError();
}

size_t scan_count_scratch_bytes() { return (size_t)kCountWorkgroups * kWalkWords * 4; }

hipError_t launch_scan_count(const ScanArgs &a, uint32_t *d_scratch, unsigned long long *d_hist, hipStream_t s)
{
    const uint64_t ngroups = (a.nblocks + kGroup - 1) / kGroup;
    if (ngroups == 0) return hipMemsetAsync(d_hist, 0, kTableWords * 8, s);
    const unsigned grid = ngroups < kCountWorkgroups ? (unsigned)ngroups : kCountWorkgroups;
    if (a.mode == 2) hipLaunchKernelGGL(scan_count_kernel<2>, dim3(grid), dim3(kGroup), 0, s, a, d_scratch, d_hist);
    else if (a.mode == 1) hipLaunchKernelGGL(scan_count_kernel<1>, dim3(grid), dim3(kGroup), 0, s, a, d_scratch, d_hist);
    else hipLaunchKernelGGL(scan_count_kernel<0>, dim3(grid), dim3(kGroup), 0, s, a, d_scratch, d_hist);
    hipLaunchKernelGGL(scan_count_sum_kernel, dim3((kWalkWords + 63) / 64, (grid + kCountRowsPerSum - 1) / kCountRowsPerSum), dim3(256), 0, s, d_scratch, grid,
                       d_hist);
    return hipGetLastError();
}

uint64_t stuff_tiles(uint64_t stream_bytes) { return (stream_bytes + kTileBytes - 1) / kTileBytes; }
uint64_t stuff_tile_bytes() { return kTileBytes; }

hipError_t launch_stuff_fused(const uint32_t *d_stream, unsigned long long *d_code_state, size_t code_state_words, uint32_t shift, bool band,
                              uint64_t max_stream_bytes, uint64_t first_tile, uint64_t tiles, unsigned long long *d_state,
                              bool state_is_zero, uint8_t *d_out, uint64_t out_cap, unsigned long long *host_totals, hipStream_t s,
                              unsigned long long *d_out_chain, uint32_t piece, const SegArgs *seg_or_null, uint32_t spin_budget)
{
    // (d_out may start anywhere: the kernel gets the 16-byte boundary below it and the distance)
    const uint32_t out_skew = (uint32_t)(reinterpret_cast<uintptr_t>(d_out) & 15);
    d_out -= out_skew;
    out_cap += out_skew;
    if (first_tile == 0 && !state_is_zero) { // (a continuation keeps the descriptors of the tiles before it)
        hipError_t e = hipMemsetAsync(d_state, 0, fused_stuff_state_words(max_stream_bytes) * 8, s);
        if (e != hipSuccess) return e;
    }
    if (tiles == 0) tiles = 1;
    if (tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (code_state_words > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const SegArgs seg = seg_or_null ? *seg_or_null : SegArgs{};
    const DispatchGate gate(s, tiles); // (held until the kernel is enqueued: dispatch_gate.hpp)
    const GateMark gm = gate.mark();
    if (seg_or_null)
        hipLaunchKernelGGL(stuff_fused_kernel<true>, dim3((unsigned)tiles), dim3(kStuffThreads), 0, s, d_stream, d_code_state, shift, band ? 1u : 0u,
                           d_state, d_out, out_skew, out_cap, first_tile, d_code_state, (uint32_t)code_state_words, host_totals, d_out_chain, piece, seg, spin_budget, gm);
    else
        hipLaunchKernelGGL(stuff_fused_kernel<false>, dim3((unsigned)tiles), dim3(kStuffThreads), 0, s, d_stream, d_code_state, shift, band ? 1u : 0u,
                           d_state, d_out, out_skew, out_cap, first_tile, d_code_state, (uint32_t)code_state_words, host_totals, d_out_chain, piece, seg, spin_budget, gm);
    return hipGetLastError();
}

} // namespace pixo_dev
