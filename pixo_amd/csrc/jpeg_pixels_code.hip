// jpeg_pixels_code.hip — pixels -> packed entropy-coded bit stream in ONE kernel (round 5): what the reference's
// baseline encode_scan does per MCU (src/jpeg/mod.rs:1448-1557: extract -> dct_2d -> quantize_block -> encode_block),
// without ever materialising the coefficient tuple.  jpeg_coeffs_kernel + scan_code_kernel wrote 3 B/px of coefficients
// to HBM and read them back; here a tile's quantised blocks go from phase B's registers straight into the flat walk.
//
// One 192-thread workgroup = one 512-pixel-wide tile = one GROUP of the scan: 32 MCUs of 4:2:0 (512x16 px) or 64 MCUs of
// 4:4:4 (512x8 px) — 192 blocks that are CONSECUTIVE in scan order (tiles of a row from left to right, rows from top to
// bottom; a row's last tile may hold fewer MCUs).  Group id g = tile_y * tiles_x + tile_x.
//
//   phase A, phase B      exactly jpeg_coeffs_kernel's (jpeg_tile.h): pixels -> planar LDS -> one lane per 8x8 block,
//                         f32 AAN rows + columns, quantiser -> the block as 32 registers of i16 pairs (natural order)
//   DC hand-off           every lane leaves its DC in LDS (index = position in scan order); the tile's LAST block of each
//                         component publishes its DC for the next tile (one 64-bit word per component and group)
//   walk                  block_pack_flat_ac (jpeg_scan_block.h; encode_block, src/jpeg/huffman.rs:438-481): the 63 AC
//                         positions + end-of-block from bit 0 of the lane's LDS scratch.  The DC symbol needs the
//                         predictor — for the tile's first block of a component that is another workgroup's value — so it
//                         is coded AFTER the walk (dc_symbol_bits) and placed in front of the AC bits when the group's bits
//                         are gathered: no lane waits for a neighbour tile before its 63-position walk is done
//   prefix                lengths scattered to scan order in LDS, three wavefront scans, exclusive prefixes gathered back
//   place + write-out     scan_code_kernel's: group aggregate published, bits OR-ed into the group's LDS window at
//                         group-relative offsets, two-level reduce-then-scan look-back (look_back_blocks), funnel-shifted
//                         coalesced stores, the word two groups share handed on through a tail descriptor
//
// The packed stream, its length (state[1], host_totals[0]) and the abort protocol are scan_code_kernel's, so the stuffing
// kernel (stuff_fused_kernel) follows unchanged.  LDS: the planar tile (16,896 B) is dead after phase B and becomes
// scratch (192 x 13 words) + window (1536 + 192 words) = 16,896 B; + 2.2 KiB of tables: 8 workgroups per CU as before.
//
// Serves one whole RGB image, 4:2:0 or 4:4:4, one uninterrupted scan with GIVEN tables (standard ones): gray images,
// restart intervals, batches, bands and optimised tables (which need the statistics of the tuple first) keep the two-kernel
// form.  Forward progress: like scan_code_kernel a group waits only for LOWER group ids (file header of
// jpeg_scan_fused.hip); every wait is bounded and raises the abort flag.
#include <hip/hip_runtime.h>

#include "jpeg_kernels.hpp"
#include "jpeg_pixels_code.hpp"
#include "jpeg_scan_dev.h"
#include "jpeg_tile.h"

#pragma clang fp contract(off)

namespace pixo_dev {
using namespace pixo_tile;

namespace {
constexpr uint64_t kDcValid = 1ull << 62;
constexpr int kFusedLds = 16896;
static_assert(kThreads == kGroup, "a tile's blocks are a group's lanes");
static_assert((kGroup * kScratchPitch + kBufWords) * 4 <= (uint32_t)kFusedLds, "scratch + window must fit the dead planar tile");
static_assert(Geo<M420>::planar <= kFusedLds && Geo<M444>::planar <= kFusedLds, "planar tile");

// what the kernel needs beyond its first (preloaded) arguments
struct PRest {
    size_t px_bytes;
    unsigned long long *clear; // housekeeping for the stuffing launch that follows (its descriptors must be zero)
    uint32_t clear_words;
    unsigned long long *host_totals;
    uint32_t spin_budget;
    uint32_t tiles_x;
    uint32_t groups;
    int16_t seed_dc[3];
    uint16_t pad_last;
};

__device__ __forceinline__ void lds_only_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Phase A of one wavefront (jpeg_kernels.hip phase_a): COUNT items of the tile, HBM -> registers -> planar LDS; all loads are
// issued before the first conversion.  Behind the pixel loads: this lane's three words of the Huffman tables in the flat
// walk's form (they arrive with the pixels and go to LDS before the barrier).
template <int MODE, int LOAD, int COUNT>
__device__ __forceinline__ void phase_a_tab(const TileCtx &c, uint32_t tx, uint32_t ty, int first, int lane, int tid, uint8_t *lds,
                                            const uint32_t *walk_tables, uint32_t *tab)
{
    typedef Geo<MODE> G;
    uint32_t r[COUNT * G::item_regs];
    const LaneAddr la = lane_addr<MODE>(c, tx, ty, lane);
#pragma unroll
    for (int j = 0; j < COUNT; j++) producer_load_item<MODE, LOAD>(c, la, tx, ty, first + j, lane, &r[j * G::item_regs]);
    uint32_t tv[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = tid + kGroup * k;
        tv[k] = walk_tables[i < kWalkWords ? i : kWalkWords - 1]; // (clamped, unconditional: no branch near a load)
    }
#pragma unroll
    for (int j = 0; j < COUNT; j++) {
        producer_fix_item<MODE, LOAD>(c, tx, first + j, lane, &r[j * G::item_regs]);
        producer_color_item<MODE, true>(first + j, lane, &r[j * G::item_regs], lds);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = tid + kGroup * k;
        if (i < kWalkWords) tab[i] = tv[k];
    }
}

template <int MODE, int LOAD, bool PACKED>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(6, 6))) void pixels_code_kernel
(const uint8_t *a_px, uint32_t a_W, uint32_t a_H, const float *a_qt, uint32_t a_units_x, uint32_t a_units_y, const uint32_t *a_tables,
 unsigned long long *a_state, uint32_t *a_stream, const PRest rest)
{
    typedef Geo<MODE> G;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kFusedLds];
    __shared__ uint32_t tab[kWalkWords];
    __shared__ uint32_t s_pos[kGroup];
    __shared__ int16_t s_dc[kGroup];
    __shared__ uint32_t wave_sum[kGroupWaves], wave_long[kGroupWaves];
    __shared__ unsigned long long s_before;
    __shared__ uint32_t s_carry, s_abort;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    __builtin_amdgcn_s_setprio(1); // phase A in front of the older workgroups' phase B (jpeg_kernels.hip)
    const uint32_t tx = blockIdx.x, ty = blockIdx.y;
    TileCtx c;
    c.px = a_px; c.y = c.cb = c.cr = nullptr; c.qt = a_qt;
    c.W = a_W; c.H = a_H; c.units_x = a_units_x; c.units_y = a_units_y; c.fast = 1;
    c.px_end = a_px + rest.px_bytes;
    if (tid == 0) { s_carry = 0; s_abort = 0; }
    {
        constexpr int base = G::items / kWaves, extra = G::items % kWaves;
        const int first = (extra && wave < extra) ? wave * (base + 1) : extra * (base + 1) + (wave - extra) * base;
        if (extra && wave < extra) phase_a_tab<MODE, LOAD, base + 1>(c, tx, ty, first, lane, tid, lds, a_tables + kTableWords, tab);
        else phase_a_tab<MODE, LOAD, base>(c, tx, ty, first, lane, tid, lds, a_tables + kTableWords, tab);
    }
    lds_only_barrier();
    __builtin_amdgcn_s_setprio(0);
    uint32_t qw[32];
    {
        float v[64];
        consumer_rows<MODE, PACKED>(wave, lane, lds, v);
        consumer_cols<PACKED>(v);
        consumer_quant<MODE, PACKED>(wave, lane, a_qt, v, qw);
    }
    // ---- from here on: the group's part of the entropy-coded scan ----------------------------------------------------------
    const uint64_t ngroups = rest.groups, g = (uint64_t)ty * rest.tiles_x + tx;
    unsigned long long *desc = a_state + 2, *tails = desc + ngroups, *dcw = tails + ngroups, *sup = dcw + 3 * ngroups;
    unsigned long long *const host_abort = rest.host_totals ? rest.host_totals + 3 : nullptr;
    // (housekeeping for the kernel that follows: its descriptors must be zero when it starts — cheaper here than a memset launch)
    for (uint64_t i = g * kGroup + tid; i < rest.clear_words; i += ngroups * kGroup) rest.clear[i] = 0;
    // which block of the scan this lane holds: MCU m of the tile, component, position among the group's 192 blocks
    const uint32_t u0 = tx * (uint32_t)G::units_x;
    const uint32_t nvalid = a_units_x - u0 < (uint32_t)G::units_x ? a_units_x - u0 : (uint32_t)G::units_x;
    uint32_t m, comp, sidx;
    bool first_of_comp, last_block_of_mcu_comp; // (the MCU's first / last block of this component)
    if (MODE == M420) {
        if (wave < 2) { m = (uint32_t)wave * 16u + ((uint32_t)lane >> 2); comp = 0; sidx = 6u * m + ((uint32_t)lane & 3u); first_of_comp = (lane & 3) == 0; last_block_of_mcu_comp = (lane & 3) == 3; }
        else { m = (uint32_t)lane & 31u; comp = 1u + ((uint32_t)lane >> 5); sidx = 6u * m + 3u + comp; first_of_comp = last_block_of_mcu_comp = true; }
    } else {
        m = (uint32_t)lane; comp = (uint32_t)wave; sidx = 3u * m + comp; first_of_comp = last_block_of_mcu_comp = true;
    }
    const bool live = m < nvalid;
    const uint32_t *wtab = tab + (comp ? kWalkClassWords : 0);
    const int dc = (int)(int16_t)(uint16_t)(qw[0] & 0xFFFFu);
    s_dc[sidx] = (int16_t)dc;
    // the tile's last block of each component: what the next tile's first block predicts from (jpeg/mod.rs:1417-1419)
    if (m + 1 == nvalid && last_block_of_mcu_comp) store_relaxed(&dcw[comp * ngroups + g], kDcValid | (uint64_t)(uint16_t)dc);
    __syncthreads(); // every wavefront has consumed its planar rows (the area becomes scratch + window), s_dc is complete
    // DC predictor: the previous block of the same component.  Inside the tile: from LDS; the tile's first block of a
    // component: the tile before (after the walk, below), or the seed for the image's first tile
    const bool external = m == 0 && first_of_comp;
    const uint32_t back = MODE == M420 ? (comp == 0 ? (first_of_comp ? 3u : 1u) : 6u) : 3u;
    int prev_dc = external ? (int)rest.seed_dc[comp] : (int)s_dc[sidx - back];
    // ---- the walk: 63 AC positions + end-of-block from bit 0 of the lane's scratch; where the packer stands is the length
    uint32_t *scratch = reinterpret_cast<uint32_t *>(lds);
    uint32_t *buf = scratch + kGroup * kScratchPitch;
    uint32_t len_ac;
    {
        FlatPack<LaneSink> p;
        p.sink = LaneSink{scratch + tid * kScratchPitch};
        p.acc = 0; p.pending = 0; p.word = 0;
        block_pack_flat_ac(qw, wtab, p);
        len_ac = p.word * 32u + p.pending;
        p.finish();
    }
    if (external && g > 0) { // (at most three lanes of the group)
        const unsigned long long *src = &dcw[comp * ngroups + g - 1];
        unsigned long long d = load_relaxed(src);
        uint32_t polls = 0;
        bool gave_up = false;
        while ((d >> 62) == 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++polls > rest.spin_budget) { gave_up = true; break; }
            d = load_relaxed(src);
        }
        if (gave_up) { raise_abort(a_state, host_abort); s_abort = 1; }
        prev_dc = (int)(int16_t)(uint16_t)(d & 0xFFFFu);
    }
    const DcBits db = dc_symbol_bits(dc, prev_dc, wtab);
    const uint32_t len = live ? db.len + len_ac : 0u;
    const bool long_block = live && len_ac > kScratchWords * 32u;
    s_pos[sidx] = len;
    __syncthreads();
    if (s_abort) return; // (a predictor never arrived: the host codes this image with the two-kernel form)
    // ---- exclusive prefix of the lengths in SCAN order: thread t takes position t
    const uint32_t mine = s_pos[tid];
    const uint32_t incl = wave_inclusive_scan(mine);
    if (lane == 63) wave_sum[wave] = incl;
    const bool any_long = PIXO_ANY64(long_block);
    if (lane == 0) wave_long[wave] = any_long ? 1u : 0u;
    __syncthreads();
    uint32_t wave_base = 0, group_bits = 0, group_long = 0;
#pragma unroll
    for (int k = 0; k < kGroupWaves; k++) {
        if (k < wave) wave_base += wave_sum[k];
        group_bits += wave_sum[k];
        group_long |= wave_long[k];
    }
    s_pos[tid] = wave_base + (incl - mine);
    if (tid == 0) publish_aggregate(desc, g, 0, group_bits);
    __syncthreads();
    const uint32_t my_bit = s_pos[sidx];
    const bool last_group = g + 1 == ngroups;
    // ---- the blocks' bits at GROUP-RELATIVE offsets into the LDS window (one round of kWindowWords words, usually), the
    // look-back, the write-out: scan_code_kernel's (jpeg_scan_fused.hip), with the DC symbol placed in front of the AC words
    const uint32_t local_words = (group_bits + 31) >> 5; // >= 1: every block has bits
    uint64_t first_word = 0;
    uint32_t sh = 0, out_words = 0, pad_word = ~0u, pad_mask = 0;
    bool tail_partial = false;
    uint32_t head_word = 0; // (thread 0) this group's bits of the stream word it shares with the group before
    for (uint32_t wbase = 0; wbase < local_words; wbase += kWindowWords) {
        const uint32_t wn = local_words - wbase < kWindowWords ? local_words - wbase : kWindowWords;
        for (uint32_t i = tid; i < wn; i += kGroup) buf[i] = 0;
        __syncthreads();
        const int64_t rel = (int64_t)my_bit - (int64_t)wbase * 32;
        const uint32_t dummy = kWindowWords + (uint32_t)tid;
        if (!group_long) {
            { // the DC symbol (<= 27 bits at the top of db.left)
                const uint32_t bsh = (uint32_t)(rel & 31), d = (uint32_t)(rel >> 5); // (wraps below zero for words before the window)
                const uint32_t hi = live ? db.left >> bsh : 0u, lo = (live && bsh) ? db.left << (32 - bsh) : 0u;
                (void)__hip_atomic_fetch_or(&buf[d < wn ? d : dummy], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_or(&buf[d + 1 < wn ? d + 1 : dummy], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // every word of the lane's scratch, shifted to its place behind the DC symbol (two LDS ORs per word)
            const int64_t rel_ac = rel + (int64_t)db.len;
            const uint32_t nw = live ? (len_ac + 31) >> 5 : 0u, bsh = (uint32_t)(rel_ac & 31);
            const uint32_t d0 = (uint32_t)(rel_ac >> 5);
#pragma unroll
            for (uint32_t j = 0; j < kScratchWords; j++) {
                if (!PIXO_ANY64(j < nw)) break; // (wave-uniform)
                const uint32_t v = j < nw ? scratch[tid * kScratchPitch + j] : 0u;
                const uint32_t d = d0 + j;
                (void)__hip_atomic_fetch_or(&buf[d < wn ? d : dummy], v >> bsh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_or(&buf[d + 1 < wn ? d + 1 : dummy], bsh ? v << (32 - bsh) : 0u, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        // (opaque to the optimiser: otherwise everything the first walk derived from the coefficients stays alive for the second)
#pragma unroll
        for (int i = 0; i < 32; i++) asm volatile("" : "+v"(qw[i]));
        if (group_long && PIXO_ANY64(live && rel < (int64_t)wn * 32 && rel + (int64_t)len > 0)) { // a block of more than 384 AC bits
            FlatPack<LdsSink> p;                                                                   // in the group: a second walk, straight into the window
            p.sink = LdsSink{buf, live ? wn : 0u, dummy};
            p.acc = 0;
            p.pending = (uint32_t)(rel & 31);
            p.word = (uint32_t)(rel >> 5);
            block_pack_flat(qw, prev_dc, wtab, p);
            p.finish();
        }
        if (wbase == 0) { // where the group starts in the stream
            if (wave == 0) {
                const uint64_t sum = look_back_blocks(desc, sup, g, 0, group_bits, a_state, host_abort, rest.spin_budget);
                if (lane == 0) {
                    if (sum == kLookBackFailed) s_abort = 1;
                    s_before = sum;
                    if (last_group) { // the stream's length in bits (unpadded)
                        a_state[1] = sum + group_bits;
                        if (rest.host_totals) rest.host_totals[0] = sum + group_bits;
                    }
                }
            }
            __syncthreads();
            if (s_abort) return;
            const uint64_t start = s_before;
            uint64_t end = start + group_bits;
            if (last_group && rest.pad_last) { // BitWriterMsb::flush pads the last byte with 1-bits
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
        // ---- out.  Stream word first_word + j = the window's words j - 1 and j funnelled by `sh`; the group writes every
        // word it completes, except the word it shares with the group before (j = 0 when sh != 0): that one waits for the
        // other group's bits.  The word it leaves unfinished goes to the next group as `tail`.
        const uint32_t carry = s_carry; // the previous window's last word
        const bool last_round = wbase + wn == local_words;
        const uint32_t upto = last_round ? out_words - wbase : wn;
        uint32_t word0 = 0, tail_word = 0;
        for (uint32_t i = tid; i < upto; i += kGroup) {
            const uint32_t j = wbase + i;
            const uint32_t cur = i < wn ? buf[i] : 0u, prev = i ? buf[i - 1] : carry;
            uint32_t v = sh ? (cur >> sh) | (prev << (32 - sh)) : cur;
            v |= j == pad_word ? pad_mask : 0u;
            const bool is_head = j == 0 && sh != 0, is_tail = tail_partial && j + 1 == out_words;
            if (is_head) word0 = v;
            if (is_tail) tail_word = v;
            if (!is_head && !is_tail) __builtin_nontemporal_store(v, &a_stream[first_word + j]);
        }
        const bool has_tail = last_round && tail_partial;
        const bool pass_through = has_tail && out_words == 1 && sh != 0; // (a handful of bits inside one word)
        if (has_tail && !pass_through && (uint32_t)tid == (upto - 1) % kGroup) store_relaxed(&tails[g], kTailValid | tail_word);
        if (wbase == 0) head_word = word0;
        if (tid == 0) s_carry = buf[wn - 1];
        __syncthreads();
    }
    // ---- the word shared with the group before: its bits arrive as that group's tail (after this group's own tail went out)
    if (sh != 0 && tid == 0) {
        uint32_t inherited = 0;
        if (g > 0) {
            unsigned long long t = load_relaxed(&tails[g - 1]);
            uint32_t polls = 0;
            while (!(t & kTailValid)) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > rest.spin_budget) { raise_abort(a_state, host_abort); return; }
                t = load_relaxed(&tails[g - 1]);
            }
            inherited = (uint32_t)t;
        }
        const uint32_t merged = inherited | head_word;
        if (tail_partial && out_words == 1) store_relaxed(&tails[g], kTailValid | merged); // (pass-through)
        else __builtin_nontemporal_store(merged, &a_stream[first_word]);
    }
}
} // namespace

size_t pixels_code_state_words(uint64_t groups)
{ // abort flag, total bits, per group: descriptor + tail + three DC words, per 64 groups: block sum (+ 1)
    return 2 + 5 * (size_t)groups + (size_t)(groups + 63) / 64 + 1;
}

uint64_t pixels_code_groups(uint32_t W, uint32_t H, bool s420)
{
    const uint32_t unit = s420 ? 16u : 8u, per_tile = s420 ? 32u : 64u;
    const uint64_t units_x = (W + unit - 1) / unit, units_y = (H + unit - 1) / unit;
    return (units_x + per_tile - 1) / per_tile * units_y;
}

bool pixels_code_supported(uint32_t W, uint32_t H, bool gray)
{ // vector pixel loads need one whole 4-pixel group per row; a tile row per grid row
    return !gray && W >= 4 && H >= 1 && (H + 7) / 8 <= 65535u;
}

hipError_t launch_pixels_code(const void *d_px, uint32_t W, uint32_t H, bool s420, const float *d_qt, const uint32_t *d_tables,
                              unsigned long long *d_state, bool state_is_zero, uint32_t *d_stream, unsigned long long *d_clear, size_t clear_words,
                              unsigned long long *host_totals, const int16_t seed_dc[3], bool pad_last, hipStream_t s, uint32_t spin_budget)
{
    if (!pixels_code_supported(W, H, false) || clear_words > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const uint32_t unit = s420 ? 16u : 8u, per_tile = s420 ? 32u : 64u;
    const uint32_t units_x = (W + unit - 1) / unit, units_y = (H + unit - 1) / unit;
    const uint32_t tiles_x = (units_x + per_tile - 1) / per_tile, tiles_y = units_y;
    const uint64_t groups = (uint64_t)tiles_x * tiles_y;
    if (groups > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (host_totals) host_totals[3] = 0; // (the kernels' abort flag)
    if (!state_is_zero) {
        const hipError_t e = hipMemsetAsync(d_state, 0, pixels_code_state_words(groups) * 8, s);
        if (e != hipSuccess) return e;
    }
    PRest rest;
    const size_t row_bytes = (size_t)W * 3;
    rest.px_bytes = row_bytes * H;
    rest.clear = d_clear; rest.clear_words = d_clear ? (uint32_t)clear_words : 0u;
    rest.host_totals = host_totals; rest.spin_budget = spin_budget;
    rest.tiles_x = tiles_x; rest.groups = (uint32_t)groups;
    for (int i = 0; i < 3; i++) rest.seed_dc[i] = seed_dc ? seed_dc[i] : (int16_t)0;
    rest.pad_last = pad_last ? 1 : 0;
    const bool aligned = reinterpret_cast<uintptr_t>(d_px) % 4 == 0 && row_bytes % 4 == 0;
    const dim3 grid(tiles_x, tiles_y);
    const uint8_t *px = static_cast<const uint8_t *>(d_px);
    const bool packed = packed_launch(groups); // (scalar or packed DCT passes and quantiser: jpeg_kernels.hpp)
#define PIXO_LAUNCH_PC(MODE, LOAD) do { if (packed) hipLaunchKernelGGL((pixels_code_kernel<MODE, LOAD, true>), grid, dim3(kThreads), 0, s, px, W, H, d_qt, units_x, units_y, d_tables, d_state, d_stream, rest); \
                                        else hipLaunchKernelGGL((pixels_code_kernel<MODE, LOAD, false>), grid, dim3(kThreads), 0, s, px, W, H, d_qt, units_x, units_y, d_tables, d_state, d_stream, rest); } while (0)
    if (s420) { if (aligned) PIXO_LAUNCH_PC(M420, L_ALIGNED); else PIXO_LAUNCH_PC(M420, L_FUNNEL); }
    else { if (aligned) PIXO_LAUNCH_PC(M444, L_ALIGNED); else PIXO_LAUNCH_PC(M444, L_FUNNEL); }
#undef PIXO_LAUNCH_PC
    return hipGetLastError();
}

} // namespace pixo_dev
