// jpeg_pixels_code.hip — pixels -> the finished, 0xFF-stuffed entropy-coded scan in ONE kernel: what the reference's
// baseline encode_scan does per MCU (src/jpeg/mod.rs:1408-1563: extract -> dct_2d -> quantize_block -> encode_block ->
// BitWriterMsb), without ever materialising the coefficient tuple or a packed bit stream in HBM.
//
// One 192-thread workgroup = one 512-pixel-wide tile = one GROUP of the scan: 32 MCUs of 4:2:0 (512x16 px) or 64 MCUs of
// 4:4:4 (512x8 px) — 192 blocks that are CONSECUTIVE in scan order (tiles of a row from left to right, rows from top to
// bottom; a row's last tile may hold fewer MCUs).  Grid (tiles_x, tiles_y, images); ticket g = (image * tiles_y + tile_y) *
// tiles_x + tile_x.
//
// GRAY images (second session of round 6): tiles of 1536 x 8 pixels = 192 consecutive blocks of one block row (phase_a_gray_row);
// OPTIMISED TABLES: the statistics come from the pixels as well (pixels_count_kernel below).
// SEGMENTS (round 6): a launch codes `images` x `segments per image` byte-aligned scans,
// each a chain of its own — DC predictors from 0, 1-padded end (BitWriterMsb::flush), look-back floor at its first group:
//   * the images of a batch (configs[2]): one segment per image, `gap` bytes left free between two scans for EOI + the next
//     file's headers (the batch then leaves the device in one copy);
//   * restart intervals that are whole MCU rows (src/jpeg/mod.rs:1423-1445 with interval = k * MCUs per row): a segment per k
//     tile rows, the two gap bytes receive FF D0+(n & 7) from the segment's last group.
// Where a segment's bytes begin: a decoupled look-back over the SEGMENTS' byte counts (+ gap), which every segment's last group
// publishes as soon as it knows its own — before it asks where its segment begins, so that the counts do not wait for each other.
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
//   place + write-out     the group's bit count published, its bits OR-ed into the group's 6 KiB LDS window at group-relative
//                         offsets, two-level reduce-then-scan look-back over the bit counts (look_back_blocks) -> the group's
//                         first bit S; its last seven bits go out for the group behind as soon as the window is complete
//   stuff + store         the group OWNS the scan's bytes [S / 8, E / 8): aligned words funnelled by S mod 8, 0xFF census,
//                         wavefront scans, the group's 0xFF count published and a second two-level look-back for the stuffed
//                         zeros before the group, bytes expanded in LDS (the gaps ARE the stuffed zeros, src/bits.rs:245-253),
//                         aligned 16-byte stores into the device buffer or straight into the caller's pinned memory
// (Round 6 built and measured ONE look-back instead of the two: a run of eight 1-bits found at every bit position of the window —
// three and-shift steps —, counted per position mod 8, every group publishing bit count + 0xFF counts for all eight alignments of
// its first bit + its first and last seven bits in 24 bytes, blocks of 64 groups their sums for all eight alignments in 40.
// Byte-identical on the whole GPU suite — and 6-7 us SLOWER per 4096x4096 file: with every group of the launch arriving at its
// look-back together, the 8 descriptor loads a lane cost more than the second round trip saves:
// profiles/r06_pixels_code_one_lookback.txt.  Also measured: the blocks' sums by device-scope ATOMIC ADDS (every group adds its
// aggregate to its block's word, count in the high bits: a block's sum complete without a reader's round trip) — identical files,
// 131 / 120 / 112 us instead of 57 / 45 / 40: 4,096 atomics on 64 words cost more than the whole kernel.  The two cheap look-backs stay.)
//
// LDS: the planar tile (16,896 B) is dead after phase B and becomes scratch (192 x 13 words) + window (1536 + 192 words)
// = 16,896 B; + 2.2 KiB of tables: 8 workgroups per CU as before.
//
// Not served: optimised tables of a batch (every file its own tables), progressive scans, restart intervals that are not whole MCU
// rows, bands of a multi-GPU image (a band's byte alignment is only known after the bit-count exchange between the GPUs: its
// stuffing cannot be fused with its coding), batches of images whose tiles are mostly empty.  Those keep
// coefficient kernel + scan_code + stuff_fused.  Forward progress: a group waits only for LOWER tickets (file header of
// jpeg_scan_fused.hip); every wait is bounded and raises the abort flag.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "dispatch_gate.hpp"
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
struct PEarly { // what phase A needs
    size_t px_bytes;           // all images
    size_t px_stride;          // bytes from one image to the next
};
// ... and what only the entropy-coding half needs.  (Loading these from the kernel-argument segment AFTER phase B instead of with the
// kernel's first instructions saves 14 of ~200 spilled scalar registers and exposes a scalar load: +0.4 .. 2 us, measured and dropped —
// profiles/r06_pixels_code_small_variants.txt.)
struct PRest {
    unsigned long long *clear; // housekeeping: the state block of the launch before this one (it must be zero when it is used again)
    uint32_t clear_words;
    unsigned long long *host_totals;
    unsigned long long *host_segs; // [segments] (pinned, or null): where every segment's bytes end in `out`
    uint32_t spin_budget;
    uint32_t tiles_x, tiles_y;
    uint32_t groups;            // all images
    uint32_t seg_rows;          // tile rows per segment (tiles_y: an image is one segment)
    uint32_t segs_per_img;
    uint32_t seg_blocks64;      // blocks of 64 groups per segment (look-back level two)
    uint32_t sup_copies, sup_stride; // the block sums exist `sup_copies` times (a power of two), `sup_stride` words apart (look_back_blocks)
    uint32_t gap;               // bytes left free between two segments' scans
    uint32_t rst;               // 1: gap == 2 and a segment's last group writes FF D0+(n & 7) there
    int16_t seed_dc[3];
    uint16_t pad_last;
    uint32_t out_skew;  // < 16: the bytes before it are somebody else's (the file headers when `out` is the caller's host buffer)
    uint64_t out_cap;   // bytes available from out[0]: nothing is stored beyond (the totals say what was needed)
    uint32_t *block_spill; // groups x 192 x 32 words: where a group of several rounds keeps its quantised blocks between the rounds' walks
    unsigned long long *gate_slots; // dispatch_gate.hpp: where the launch's last eight workgroups say that they have started
    unsigned long long gate_seq;
};

// A value every lane of the wavefront holds alike — read from LDS, say — as the compiler can SEE it: a scalar register.  Branches
// on it become real branches (a register that is dead on one side is free there) instead of two exec-masked regions in a row.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }

__device__ __forceinline__ void lds_only_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The lane's scratch as the first walk's sink WITHOUT the clamp of LaneSink (jpeg_scan_dev.h): a block longer than its scratch writes
// on into LDS that belongs to nobody who will read it (see the walk below) — one v_min per position less.  A block has at most 1,665
// bits = 53 words: the last lane's reach ends inside the window.
static_assert((kGroup - 1) * kScratchPitch + 54 <= kGroup * kScratchPitch + kBufWords, "a long block's overrun stays inside the LDS area");
struct LaneSinkOpen {
    uint32_t *words;
    __device__ __forceinline__ void or_word(bool, uint32_t word, uint32_t value) { words[word] = value; }
};

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

// GRAY images (second session of round 6): the coefficient kernel's gray tile is 512 x 24 pixels — three block rows, not a run of the
// scan order.  The fused kernels take a tile of 1536 x 8 pixels instead: 192 CONSECUTIVE blocks of one block row, wavefront w the 64
// blocks [64 w, 64 w + 64) — in LDS exactly the gray planes phase B of jpeg_tile.h reads (plane w, rows 0..7, pitch kPitch), so only
// the addresses of phase A differ: item k (48 of 256 bytes each, 16 per wavefront) = row k / 6, sixth k % 6 of the tile row.
// extract_block's replicate rule (src/jpeg/mod.rs:1565-1606): rows clamped to H - 1, the row's last 4-pixel group read at W - 4 and
// permuted like producer_fix_item's gray case.
constexpr int kGrayTileBlocks = 192;
template <int MODE> constexpr int tile_units() { return MODE == MGRAY ? kGrayTileBlocks : Geo<MODE>::units_x; }
template <int LOAD>
__device__ __forceinline__ void phase_a_gray_row(const TileCtx &c, uint32_t tx, uint32_t ty, int wave, int lane, uint8_t *lds)
{
    const uint32_t last = c.W - 4u; // (W >= 4: pixels_code_supported)
    uint32_t r[16], d[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int k = wave * 16 + j, row = k / 6, sixth = k % 6;
        const uint32_t y0 = ty * 8u + (uint32_t)row, y = y0 < c.H ? y0 : c.H - 1u;
        const uint32_t x = tx * 1536u + (uint32_t)sixth * 256u + 4u * (uint32_t)lane, xc = x < last ? x : last;
        const uint8_t *p = c.px + (size_t)y * c.W;
        if (LOAD == L_ALIGNED) r[j] = PIXO_GLOAD((const uint32_t *)(p + xc));
        else unaligned_load<1>(p, xc, &r[j]);
        const uint32_t over = x > last ? x - last : 0u;
        d[j] = over < 3u ? over : 3u;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int k = wave * 16 + j, row = k / 6, sixth = k % 6;
        const uint32_t s1 = perm(r[j], r[j], 0x03030201u), s2 = perm(r[j], r[j], 0x03030302u), s3 = perm(r[j], r[j], 0x03030303u);
        const uint32_t v = d[j] == 0 ? r[j] : (d[j] == 1 ? s1 : (d[j] == 2 ? s2 : s3));
        *(uint32_t *)(lds + (sixth >> 1) * 4224 + row * kPitch + 4 * ((sixth & 1) * 64 + lane)) = v;
    }
}

#ifdef PIXO_TIMELINE // (experiment builds only, tools/pixels_code_timeline.py: where a group's time goes; 100 MHz constant clock)
__device__ unsigned long long g_pc_timeline[8192 * 16];
#define PIXO_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 8192) g_pc_timeline[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PIXO_STAMP(k) do { } while (0)
#endif

// SEG = false: ONE image, ONE segment — the shape of the metric's whole-file path: everything about segments folds away (the
// kernel of round 5: the segment bookkeeping costs ~1 us of a 40-57 us launch in uniform arithmetic and spilled scalar registers).
template <int MODE, int LOAD, bool PACKED, bool SEG>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(6, 6))) void pixels_code_kernel
(const uint8_t *a_px, uint32_t a_W, uint32_t a_H, const float *a_qt, uint32_t a_units_x, uint32_t a_units_y, const uint32_t *a_tables,
 unsigned long long *a_state, uint8_t *a_out, const PEarly early, const PRest rest_by_value)
{
    typedef Geo<MODE> G;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kFusedLds];
    __shared__ uint32_t tab[kWalkWords];
    __shared__ uint32_t s_pos[kGroup];
    __shared__ int16_t s_dc[kGroup];
    __shared__ uint32_t wave_sum[kGroupWaves], wave_long[kGroupWaves];
    __shared__ unsigned long long s_before, s_segbase;
    __shared__ uint32_t s_carry, s_abort, s_head, s_first_ones;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    PIXO_STAMP(0);
    __builtin_amdgcn_s_setprio(1); // phase A in front of the older workgroups' phase B (jpeg_kernels.hip)
    const uint32_t tx = blockIdx.x, ty = blockIdx.y, img = SEG ? blockIdx.z : 0u;
    TileCtx c;
    c.px = a_px + (size_t)img * early.px_stride; c.y = c.cb = c.cr = nullptr; c.qt = a_qt;
    c.W = a_W; c.H = a_H; c.units_x = a_units_x; c.units_y = a_units_y; c.fast = 1;
    c.px_end = a_px + early.px_bytes;
    if (tid == 0) { s_carry = 0; s_abort = 0; s_head = 0; s_first_ones = 0; s_segbase = 0; }
    if (!SEG) {
        // dispatch_gate.hpp: the launch's last eight workgroups say that they have started — AT their start (behind phase A it was 8 us
        // later, which the next thread's launch spent waiting), and found from the preloaded arguments alone: no workgroup waits for
        // the rest of the kernel-argument segment because of this (a batch, whose grid has a third dimension, marks behind phase A)
        const uint32_t tiles_x = (a_units_x + (uint32_t)tile_units<MODE>() - 1u) / (uint32_t)tile_units<MODE>(), total = tiles_x * a_units_y, lin = blockIdx.y * tiles_x + blockIdx.x;
        if (lin + 8u >= total || lin == 0u) {
            if (tid == 0) dispatch_mark(rest_by_value.gate_slots, rest_by_value.gate_seq, lin, total);
        }
    }
    if constexpr (MODE == MGRAY) {
        phase_a_gray_row<LOAD>(c, tx, ty, wave, lane, lds);
#pragma unroll
        for (int k = 0; k < 3; k++) { // (the flat walk's tables: phase_a_tab's part)
            const int i = tid + kGroup * k;
            if (i < kWalkWords) tab[i] = a_tables[kTableWords + i];
        }
    } else {
        constexpr int base = G::items / kWaves, extra = G::items % kWaves;
        const int first = (extra && wave < extra) ? wave * (base + 1) : extra * (base + 1) + (wave - extra) * base;
        if (extra && wave < extra) phase_a_tab<MODE, LOAD, base + 1>(c, tx, ty, first, lane, tid, lds, a_tables + kTableWords, tab);
        else phase_a_tab<MODE, LOAD, base>(c, tx, ty, first, lane, tid, lds, a_tables + kTableWords, tab);
    }
    lds_only_barrier();
    PIXO_STAMP(1);
    __builtin_amdgcn_s_setprio(0);
    if (SEG && tid == 0) dispatch_mark(rest_by_value.gate_slots, rest_by_value.gate_seq, ((uint64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x,
                                       (uint64_t)gridDim.x * gridDim.y * gridDim.z);
    uint32_t qw[32];
    {
        float v[64];
        consumer_rows<MODE, PACKED>(wave, lane, lds, v);
        consumer_cols<PACKED>(v);
        consumer_quant<MODE, PACKED>(wave, lane, a_qt, v, qw);
    }
    PIXO_STAMP(2);
    // ---- from here on: the group's part of the entropy-coded scan ----------------------------------------------------------
    // (the thread's coordinates again, opaque to the optimiser: whatever the second half derives from them is computed here and
    // does not occupy registers through phase B, which runs at the 80-register limit)
    int tid_again = threadIdx.x;
    asm volatile("" : "+v"(tid_again));
#define tid tid_again
    const int lane_again = tid & 63, wave_again = __builtin_amdgcn_readfirstlane(tid >> 6);
#define lane lane_again
#define wave wave_again
    const PRest &rest = rest_by_value;
    // ticket, segment, place in the segment's chain
    const uint32_t tiles_x = rest.tiles_x, tiles_y = rest.tiles_y;
    const uint64_t ngroups = rest.groups, g = ((uint64_t)img * tiles_y + ty) * tiles_x + tx;
    const uint32_t srow = (!SEG || rest.seg_rows >= tiles_y) ? 0u : ty / rest.seg_rows;       // (wave-uniform)
    const uint32_t seg = SEG ? img * rest.segs_per_img + srow : 0u, nsegs = SEG ? gridDim.z * rest.segs_per_img : 1u;
    const uint32_t row0 = SEG ? srow * rest.seg_rows : 0u;
    const uint32_t rows_here = (!SEG || tiles_y - row0 < rest.seg_rows) ? tiles_y - row0 : rest.seg_rows;
    const uint32_t rel = (ty - row0) * tiles_x + tx, seg_groups = rows_here * tiles_x;
    const uint64_t floor = g - rel;
    const bool last_group = rel + 1 == seg_groups; // of its segment
    // state: [0] abort flag, [1] total bits; per group: bit-count descriptor, tail, three DC words, 0xFF-count descriptor; per
    // segment: two rows of block sums (one per 64 groups, + 1) and the segment's byte count / the offset of the next segment
    const uint32_t sup_copies = rest.sup_copies, sup_stride = rest.sup_stride;
    const uint64_t sup_words = (uint64_t)sup_copies * sup_stride;
    unsigned long long *desc = a_state + 2, *tails = desc + ngroups, *dcw = tails + ngroups, *desc2 = dcw + 3 * ngroups, *sups = desc2 + ngroups,
                       *sups2 = sups + sup_words, *segdesc = sups2 + sup_words;
    unsigned long long *SUP = sups + (uint64_t)seg * rest.seg_blocks64, *SUP2 = sups2 + (uint64_t)seg * rest.seg_blocks64; // (copy 0)
    unsigned long long *const host_abort = rest.host_totals ? rest.host_totals + 3 : nullptr;
    // (housekeeping: the state block of the launch before this one must be zero when it is used again — cheaper here than a memset launch)
    for (uint64_t i = g * kGroup + tid; i < rest.clear_words; i += ngroups * kGroup) rest.clear[i] = 0;
    // which block of the scan this lane holds: MCU m of the tile, component, position among the group's 192 blocks
    const uint32_t u0 = tx * (uint32_t)tile_units<MODE>();
    const uint32_t nvalid = a_units_x - u0 < (uint32_t)tile_units<MODE>() ? a_units_x - u0 : (uint32_t)tile_units<MODE>();
    uint32_t m, comp, sidx;
    bool first_of_comp, last_block_of_mcu_comp; // (the MCU's first / last block of this component)
    if (MODE == MGRAY) { // one component, an MCU is a block: the tile's 192 blocks in scan order
        m = (uint32_t)tid; comp = 0; sidx = m; first_of_comp = last_block_of_mcu_comp = true;
    } else if (MODE == M420) {
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
    if (m + 1 == nvalid && last_block_of_mcu_comp && !last_group) store_relaxed(&dcw[comp * ngroups + g], kDcValid | (uint64_t)(uint16_t)dc);
    __syncthreads(); // every wavefront has consumed its planar rows (the area becomes scratch + window), s_dc is complete
    // DC predictor: the previous block of the same component.  Inside the tile: from LDS; the tile's first block of a
    // component: the tile before (after the walk, below); a segment's first tile: zero (the launch's first: the seed)
    const bool external = m == 0 && first_of_comp;
    const uint32_t back = MODE == MGRAY ? 1u : (MODE == M420 ? (comp == 0 ? (first_of_comp ? 3u : 1u) : 6u) : 3u);
    int prev_dc = external ? (seg == 0 ? (int)rest.seed_dc[comp] : 0) : (int)s_dc[sidx - back];
    uint32_t *scratch = reinterpret_cast<uint32_t *>(lds);
    uint32_t *buf = scratch + kGroup * kScratchPitch;
    // the window, zero: the gather ORs into it (behind the next barrier; the walk only touches the scratch in front of it)
    static_assert(kWindowWords % (4 * kGroup) == 0 && (kGroup * kScratchPitch) % 4 == 0, "the window in whole 16-byte pieces per lane");
#pragma unroll
    for (uint32_t i = 0; i < kWindowWords / (4 * kGroup); i++) reinterpret_cast<v4u *>(buf)[(uint32_t)tid + kGroup * i] = v4u{0, 0, 0, 0};
    // ---- the walk: 63 AC positions + end-of-block from bit 0 of the lane's scratch; where the packer stands is the length
    uint32_t len_ac;
    {
        // (the block in U-FORM from here on — u = v - (v < 0), jpeg_scan_block.h — in place: the walk, the parked copy of a long group
        // and its second walks all take that form; a block of more than kScratchWords x 32 bits runs over into the scratch of the lanes
        // behind it and, for the tile's last lanes, into the window: such a group takes the long-block path, which never reads the
        // scratch and zeroes the window again)
        block_to_u(qw);
        FlatPack<LaneSinkOpen> p;
        p.sink = LaneSinkOpen{scratch + tid * kScratchPitch};
        p.acc = 0; p.pending = 0; p.word = 0;
        block_pack_flat_ac_u(qw, wtab, p);
        len_ac = p.word * 32u + p.pending;
        p.finish();
    }
    PIXO_STAMP(3);
    if (external && rel > 0) { // (at most three lanes of the group)
        const unsigned long long *src = &dcw[comp * ngroups + g - 1];
        unsigned long long d = load_relaxed(src);
        uint32_t polls = 0;
        bool gave_up = false;
        while ((d >> 62) == 0) {
            __builtin_amdgcn_s_sleep(kPollSleep);
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
    if (uni(s_abort)) return; // (a predictor never arrived: the host codes this image with the two-kernel form)
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
    group_bits = uni(group_bits); group_long = uni(group_long);
    s_pos[tid] = wave_base + (incl - mine);
    if (tid == 0) publish_aggregate(desc, g, floor, group_bits);
    __syncthreads();
    const uint32_t my_bit = s_pos[sidx];
    PIXO_STAMP(4);
    // ---- place, stuff, write.  The group's bits go into the LDS window at GROUP-RELATIVE offsets (scan_code_kernel's gather, with
    // the DC symbol in front of the AC words); a look-back over the groups' bit counts gives the group's first bit S in its
    // scan; from there on everything is BYTES of the finished scan (stuff_fused_kernel's work happens here, the packed stream
    // never exists in HBM):
    //   * the group OWNS the scan's bytes [S / 8, E / 8) (E = S + its bits; a segment's last group: up to the 1-padded end).  A first
    //     byte that began in the group before (S % 8 != 0) arrives as that group's `tail` — its last E' % 8 bits, which it
    //     publishes as soon as its window is complete, before it waits for anything — and is completed here;
    //   * "aligned word" j of the group = its owned bytes 4 j .. 4 j + 3 = the window's words j - 1 and j funnelled by S % 8;
    //     thread (wave v, row k, lane l) takes aligned word 512 v + 64 k + l of the round — 0xFF census per word, wavefront
    //     scans row by row, the group's count published and a second look-back (both two-level, look_back_blocks' form) for
    //     the number of stuffed zeros before the group;
    //   * the bytes are expanded into an LDS stage at the output's 16-byte alignment — each moved up by the 0xFF bytes before
    //     it: the gaps ARE the stuffed zeros (BitWriterMsb, src/bits.rs:245-253) — and stored as aligned 16-byte pieces.
    // (Round 6 measured ONE look-back instead — every group's 0xFF counts for all eight alignments of its first bit in a 24-byte
    // descriptor, a predecessor's alignment from the running bit sum: byte-identical, and 6-7 us slower per 4096x4096 file: the
    // 8 descriptor loads a lane cost more than the second round trip saves, profiles/r06_pixels_code_one_lookback.txt.)
    // The stage is the walk's scratch area (dead after the gather): a group of several rounds, whose later gathers would need the
    // scratch again, packs every round by a second walk straight into the window (the long-block path).
    constexpr uint32_t kWin = kWindowWords - 1; // group words per round: the round's aligned words (one more) are at most 3 x 512
    constexpr uint32_t kStageCap = kGroup * kScratchPitch * 4;
    constexpr int kRows = 8;                    // aligned words per lane and round
    static_assert(kWindowWords == 3 * 64 * kRows, "three wavefronts x 8 rows of 64 aligned words");
    const uint32_t local_words = (group_bits + 31) >> 5; // >= 1: every block has bits
    const bool walk_into_window = group_long != 0 || local_words > kWin;
    uint8_t *stage = lds;
    uint8_t *const out = a_out; // 16-byte aligned; the launch's first scan byte goes to out[rest.out_skew]
    uint64_t S = 0, ff_before_groups = 0, seg_base = 0;
    uint32_t nb_total = 0, sh8 = 0, pad_word = ~0u, pad_mask = 0, ff_group = 0;
    bool aborted = false;
    // MULTI: a group of several rounds / with a very long block (rare: noise at q >= 90): the quantised block stays alive for the
    // second walks.  The common case is its own instantiation, in which the block's registers are dead after the first walk.
    // (such a group parks its blocks in HBM — a slot per lane — and loads them again for every round's walk: the 32 registers are
    // not alive through the byte stage of the common path)
    uint32_t *const my_spill = rest.block_spill + ((size_t)g * kGroup + (size_t)tid) * 32u;
    uint32_t park = uni(walk_into_window ? 1u : 0u);
    asm volatile("" : "+s"(park)); // (not recognisable as the condition of the branch below: otherwise this store is moved into that branch
                                   //  and the 32 registers stay occupied through the other side's code as well)
    if (park) {
#pragma unroll
        for (int i = 0; i < 8; i++) *reinterpret_cast<v4u *>(my_spill + 4 * i) = v4u{qw[4 * i], qw[4 * i + 1], qw[4 * i + 2], qw[4 * i + 3]};
    }
    // A segment's LAST group publishes the segment's byte count before it asks where the segment begins (the segments' counts then
    // do not wait for each other); a last group of several rounds knows its count only at the end and asks first, like everybody else.
    const bool seg_late = last_group && !park;
    const uint32_t kblk = rel >> 6, in_block = rel & 63u; // the group's block of 64 inside its chain, its place in the block
    auto rounds = [&](auto multi_tag) __attribute__((always_inline)) {
    constexpr bool MULTI = decltype(multi_tag)::value;
    // The second look-back — the number of stuffed zeros before the group — as a step of its own: own_ff = ALL the group's 0xFF bytes (its
    // count has gone out; the block's sum goes out in here, before the wait for the other blocks' sums).  false: gave up (aborted).
    auto second_look_back = [&](uint32_t own_ff) __attribute__((always_inline)) -> bool {
        if (wave == 0) { // (look_back_blocks' two levels; the block's sum is published below, when this group's own count is complete)
            int lane2 = lane; // (a value of this place: see look_back_blocks)
            asm volatile("" : "+v"(lane2));
#undef lane
#define lane lane2
            const uint64_t block_first = floor + ((uint64_t)kblk << 6);
            uint32_t polls = 0;
            bool gave_up = false;
            unsigned long long *const SUP2r = SUP2 + (size_t)((uint32_t)g & (sup_copies - 1u)) * sup_stride; // the copy this group reads
            unsigned long long da = (uint32_t)lane < in_block ? load_relaxed(&desc2[block_first + lane]) : kFlagAggregate;
            unsigned long long db2 = (uint32_t)lane < kblk ? load_relaxed(&SUP2r[lane]) : kFlagAggregate;
            while ((da >> 62) == 0 && !gave_up) {
                __builtin_amdgcn_s_sleep(kPollSleep);
                if (++polls > rest.spin_budget) gave_up = true; else da = load_relaxed(&desc2[block_first + lane]);
            }
            uint64_t before2 = 0;
            uint32_t front = 0;
            if (!PIXO_ANY64(gave_up)) {
                front = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan((uint32_t)(da & kValueMask)), 63);
                before2 = front;
                // the block's own sum goes out BEFORE waiting for the other blocks' sums (a group of one round knows its count here):
                // published behind that wait, the blocks' last groups would form one chain of waits through the whole scan
                if (in_block == 63u && (uint32_t)lane < sup_copies) store_relaxed(&SUP2[(size_t)lane * sup_stride + kblk], kFlagAggregate | ((uint64_t)front + own_ff));
                for (uint32_t base = 0;;) {
                    while ((db2 >> 62) == 0 && !gave_up) {
                        __builtin_amdgcn_s_sleep(kPollSleep);
                        if (++polls > rest.spin_budget) gave_up = true; else db2 = load_relaxed(&SUP2r[base + lane]);
                    }
                    if (PIXO_ANY64(gave_up)) break;
                    before2 += (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan((uint32_t)(db2 & kValueMask)), 63);
                    base += 64;
                    if (base >= kblk) break;
                    db2 = base + lane < kblk ? load_relaxed(&SUP2r[base + lane]) : kFlagAggregate;
                }
            }
            if (PIXO_ANY64(gave_up)) {
                if (gave_up) raise_abort(a_state, host_abort);
                if (lane == 0) s_abort = 1;
            } else if (lane == 0) {
                s_before = before2;
            }
#undef lane
#define lane lane_again
        } else if (SEG && wave == 1 && seg > 0 && !seg_late) {
            // meanwhile: where this segment's bytes begin — a look-back over the SEGMENTS' byte counts.  HERE, behind the group's own
            // 0xFF count going out, not beside the first look-back: the count of the segment before is known when ITS last group is
            // through its second look-back, and a group that waited for that in front of its census kept its own count from the
            // groups behind it — every segment's last group then finished one census + one look-back (6.5 us) behind the last group
            // of the segment before: a serial chain through all segments (64 x 1080p: 414-453 us; 32 x 4096x512: 216-236 us
            // against 126-141 us for the same groups in 4 segments, profiles/r06_fused_batches_chain.txt).
            const uint64_t sb = look_back(segdesc, seg, 0, 0, a_state, host_abort, rest.spin_budget);
            if (lane == 0) {
                if (sb == kLookBackFailed) s_abort = 1;
                s_segbase = sb;
            }
        }
        __syncthreads();
        PIXO_STAMP(8);
        if (uni(s_abort)) { aborted = true; return false; }
        ff_before_groups = uni64(s_before);
        if (SEG) seg_base = uni64(s_segbase);
        return true;
    };
    for (int pass = 0; pass < (MULTI ? 2 : 1); pass++) {
    if (MULTI && pass == 1) { // the count is complete: out it goes, then the look-back, then the rounds again — this time for their bytes
        if (rel > 0 && sh8 != 0) { // (uniform) the group starts inside a byte: the seven bits in front of it, and whether that byte is 0xFF
            if (tid == 128) {
                unsigned long long t = load_relaxed(&tails[g - 1]);
                uint32_t polls = 0;
                while (!(t & kTailValid)) {
                    __builtin_amdgcn_s_sleep(kPollSleep);
                    if (++polls > rest.spin_budget) { raise_abort(a_state, host_abort); s_abort = 1; break; }
                    t = load_relaxed(&tails[g - 1]);
                }
                s_head = (uint32_t)t & 0x7Fu;
            }
            __syncthreads();
            if (uni(s_abort)) { aborted = true; return; }
            const uint32_t mask = (1u << sh8) - 1u;
            if (uni(s_first_ones) && (uni(s_head) & mask) == mask) ff_group += 1u;
        }
        if (tid == 0) store_relaxed(&desc2[g], kFlagAggregate | (uint64_t)ff_group);
        if (!second_look_back(ff_group)) return;
        ff_group = 0;
    }
    for (uint32_t wbase = 0; wbase < (MULTI ? local_words : 1u); wbase += kWin) {
        const uint32_t wn = MULTI ? (local_words - wbase < kWin ? local_words - wbase : kWin) : local_words;
        if (MULTI) { // (also the first round's: the walk of a long block ran over its scratch into the window)
#pragma unroll
            for (uint32_t i = 0; i < kWindowWords / kGroup; i++) buf[(uint32_t)tid + kGroup * i] = 0;
            __syncthreads();
        }
        const int64_t rel_bit = (int64_t)my_bit - (int64_t)wbase * 32;
        const uint32_t dummy = kWindowWords + (uint32_t)tid;
        if (!MULTI) {
            // (No clamping of the word indices: a word behind the group's last one only ever receives ZERO — a block's bits end where
            // its length says, and the words of a lane without a block are zero — and d + 1 <= local_words + kScratchWords + 1 lies
            // inside the window + the dummies.  {v, 0} >> s = the part of v that moves into the next word, 0 for s = 0: one alignbit.)
            static_assert(kWindowWords + kScratchWords + 2 <= kBufWords, "the gather's last word index stays inside the buffer");
            { // the DC symbol (<= 27 bits at the top of db.left)
                const uint32_t bsh = (uint32_t)(rel_bit & 31), d = (uint32_t)(rel_bit >> 5);
                const uint32_t left = live ? db.left : 0u;
                (void)__hip_atomic_fetch_or(&buf[d], left >> bsh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_or(&buf[d + 1], __builtin_amdgcn_alignbit(left, 0u, bsh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // every word of the lane's scratch, shifted to its place behind the DC symbol (two LDS ORs per word)
            const int64_t rel_ac = rel_bit + (int64_t)db.len;
            const uint32_t nw = live ? (len_ac + 31) >> 5 : 0u, bsh = (uint32_t)(rel_ac & 31);
            const uint32_t d0 = live ? (uint32_t)(rel_ac >> 5) : 0u;
#pragma unroll
            for (uint32_t j = 0; j < kScratchWords; j++) {
                if (!PIXO_ANY64(j < nw)) break; // (wave-uniform)
                const uint32_t v = j < nw ? scratch[tid * kScratchPitch + j] : 0u;
                const uint32_t d = d0 + j;
                (void)__hip_atomic_fetch_or(&buf[d], v >> bsh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_or(&buf[d + 1], __builtin_amdgcn_alignbit(v, 0u, bsh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (MULTI && PIXO_ANY64(live && rel_bit < (int64_t)wn * 32 && rel_bit + (int64_t)len > 0)) { // (some block of the wavefront lies in the window)
            uint32_t w[32]; // the lane's block again (its own stores: visible to it)
            const uint32_t *back_ptr = my_spill;
            asm volatile("" : "+v"(back_ptr) : : "memory"); // (a pointer the optimiser cannot see through: no forwarding of the stored registers, which would keep them alive)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const v4u q = *reinterpret_cast<const v4u *>(back_ptr + 4 * i);
                w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
            }
            FlatPack<LdsSink> p;
            p.sink = LdsSink{buf, live ? wn : 0u, dummy};
            p.acc = 0;
            p.pending = (uint32_t)(rel_bit & 31);
            p.word = (uint32_t)(rel_bit >> 5);
            block_pack_flat_u(w, prev_dc, wtab, p); // (parked in u-form)
            p.finish();
        }
        const bool last_round = !MULTI || wbase + wn == local_words;
        __syncthreads(); // the round's window is complete; the scratch is dead (it becomes the stage)
        PIXO_STAMP(5);
        // The group's LAST SEVEN BITS for the group behind (its first byte may begin in this group): they do not depend on where
        // this group starts, so they go out at once — the group behind then finds them waiting instead of waiting for them.
        // (>= 12 bits per group: they are this group's own.)
        // (A group of FEWER than seven bits — one block of a narrow gray image, one MCU under optimised tables with 1-bit codes — has no seven
        // bits of its own: it first asks for the seven in front of it and hands on theirs + its own, behind the look-back below.)
        if (last_round && !last_group && tid == 64 && pass == 0 && (group_bits >= 7u || rel == 0)) {
            const uint32_t e = group_bits - wbase * 32u; // the end of the group's bits in this window
            const uint32_t wi = (e - 1u) >> 5, lo = wi ? buf[wi - 1] : (wbase ? s_carry : 0u), hi = buf[wi];
            const uint32_t used = e - wi * 32u; // bits of word wi in use (1..32): the last 7 bits = bits [used - 7, used) of {lo, hi}
            const uint64_t both = ((uint64_t)lo << 32) | hi;
            store_relaxed(&tails[g], kTailValid | (uint32_t)((both >> (32u - used)) & 0x7Fu));
        }
        if (wbase == 0 && pass == 0) { // where the group starts in its scan
            if (wave == 0) {
                const uint64_t sum = look_back_blocks(desc, SUP, g, floor, group_bits, a_state, host_abort, rest.spin_budget, sup_copies, sup_stride);
                if (lane == 0) {
                    if (sum == kLookBackFailed) s_abort = 1;
                    s_before = sum;
                }
            } else if (!MULTI && tid == 128 && rel > 0) { // meanwhile: the seven bits in front of this group (used if it starts inside a byte)
                // (a group of several rounds asks for them BEHIND its count pass — below: it lets its own last seven bits out in that pass's
                // last round, and asking first made every such group wait for the whole count pass of the group before it)
                unsigned long long t = load_relaxed(&tails[g - 1]);
                uint32_t polls = 0;
                while (!(t & kTailValid)) {
                    __builtin_amdgcn_s_sleep(kPollSleep);
                    if (++polls > rest.spin_budget) { raise_abort(a_state, host_abort); s_abort = 1; break; }
                    t = load_relaxed(&tails[g - 1]);
                }
                s_head = (uint32_t)t & 0x7Fu;
            }
            // (the stage, zero — the scratch area is dead: a round that expands all its bytes at once needs no zeroing pass of its own.
            // As far as this round can need it: its bytes, as many stuffed zeros, the 16-byte alignment at both ends)
            {
                const uint32_t need = 2u * ((wn < kWin ? wn : kWin) * 4u + 8u) + 32u, upto = need < kStageCap ? need : kStageCap;
                for (uint32_t i = 16u * tid; i < upto; i += 16u * kGroup) *reinterpret_cast<v4u *>(stage + i) = v4u{0, 0, 0, 0};
            }
            __syncthreads();
            PIXO_STAMP(6);
            if (uni(s_abort)) { aborted = true; return; }
            if (!MULTI && group_bits < 7u && rel > 0 && !last_group && tid == 64) // (see above; every block has at least two bits)
                store_relaxed(&tails[g], kTailValid | (((s_head << group_bits) | (buf[0] >> (32u - group_bits))) & 0x7Fu));
            S = uni64(s_before);
            sh8 = (uint32_t)(S & 7);
            const uint64_t end_bits = (uint64_t)sh8 + group_bits; // in aligned bits: bit 0 = the first bit of the group's first owned byte
            const bool pad = last_group && rest.pad_last;
            nb_total = (uint32_t)(pad ? (end_bits + 7) >> 3 : end_bits >> 3);
            if (pad && (end_bits & 7)) { // BitWriterMsb::flush pads the last byte with 1-bits
                const uint32_t n = 8u - (uint32_t)(end_bits & 7);
                pad_word = (uint32_t)(end_bits >> 5);
                pad_mask = ((1u << n) - 1u) << (32u - (uint32_t)(end_bits & 31) - n);
            }
        } else {
            for (uint32_t i = 16u * tid; i < kStageCap; i += 16u * kGroup) *reinterpret_cast<v4u *>(stage + i) = v4u{0, 0, 0, 0};
            __syncthreads();
        }
        // ---- the round's aligned words: thread (wave, row k, lane) takes word 512 wave + 64 k + lane
        const uint32_t head = uni(wbase ? s_carry : 0u); // a later round: the previous round's last word is in front of its first aligned word
        uint32_t x[kRows];
        auto aligned_word = [&](uint32_t jl, uint32_t first_prev) -> uint32_t {
            // (the window is zero behind the round's words — it was zeroed whole — so no index needs clamping; {prev, cur} >> S % 8)
            const uint32_t cur = buf[jl], prev = jl ? buf[jl - 1] : first_prev;
            uint32_t v = __builtin_amdgcn_alignbit(prev, cur, sh8);
            v |= wbase + jl == pad_word ? pad_mask : 0u;
            return v;
        };
        // the last sh8 of the seven bits in front (a count pass runs without them: its first byte then cannot be 0xFF, and whether it is
        // is settled behind the pass — s_first_ones: the group's own part of that byte is all ones)
        const uint32_t head_now = wbase ? head : ((MULTI && pass == 0) ? 0u : (uni(s_head) & ((1u << sh8) - 1u)));
        // owned bytes of THIS round: up to the group's last one, or (not the last round) up to the round's last complete aligned word
        const uint32_t round_first = 4u * wbase;
        const uint32_t limit = last_round ? nb_total : (nb_total < 4u * (wbase + wn) ? nb_total : 4u * (wbase + wn));
        const uint32_t bytes_this = limit > round_first ? limit - round_first : 0u;
        uint64_t flags = 0; // four flags per word: byte b of word k is 0xFF and belongs to this round
        const uint32_t jl0 = 512u * (uint32_t)wave + (uint32_t)lane;
#pragma unroll
        for (int k = 0; k < kRows; k++) {
            x[k] = 0;
            if (4u * (wbase + 512u * (uint32_t)wave + 64u * k) >= limit) continue; // (wave-uniform: the row lies behind the round's bytes — most rows of a smooth image's groups)
            const uint32_t jl = jl0 + 64u * k;
            x[k] = aligned_word(jl, head_now);
            const uint32_t first_byte = 4u * (wbase + jl);
            const uint32_t exist = first_byte < limit ? (limit - first_byte < 4u ? limit - first_byte : 4u) : 0u;
            const uint32_t mm = (zero_byte_mask(~x[k]) >> 7) & 0x01010101u; // bits 24, 16, 8, 0 = bytes 0, 1, 2, 3 of the word (stream order)
            const uint32_t m4 = ((mm * 0x08040201u) >> 24) & ((1u << exist) - 1u);
            flags |= (uint64_t)m4 << (4 * k);
        }
        // 0xFF bytes before every word of the wavefront, in stream order = row by row; two rows share one 32-bit scan
        uint32_t before[kRows];
        uint32_t wave_ff = 0;
#pragma unroll
        for (int k = 0; k < kRows; k += 2) {
            if (4u * (wbase + 512u * (uint32_t)wave + 64u * k) >= limit) { before[k] = before[k + 1] = wave_ff; continue; } // (no bytes, no 0xFF bytes)
            const uint32_t c0 = (uint32_t)__builtin_popcount((uint32_t)(flags >> (4 * k)) & 0xFu);
            const uint32_t c1 = (uint32_t)__builtin_popcount((uint32_t)(flags >> (4 * k + 4)) & 0xFu);
            const uint32_t sc = wave_inclusive_scan(c0 | (c1 << 16));
            const uint32_t rows = (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
            before[k] = wave_ff + (sc & 0xFFFFu) - c0;
            wave_ff += rows & 0xFFFFu;
            before[k + 1] = wave_ff + (sc >> 16) - c1;
            wave_ff += rows >> 16;
        }
        if (lane == 0) wave_sum[wave] = wave_ff;
        __syncthreads();
        PIXO_STAMP(7);
        uint32_t wave_base_ff = 0, round_ff = 0;
        uint32_t ff_of_wave[kGroupWaves];
#pragma unroll
        for (int k = 0; k < kGroupWaves; k++) {
            ff_of_wave[k] = uni(wave_sum[k]);
            if (k < wave) wave_base_ff += ff_of_wave[k];
            round_ff += ff_of_wave[k];
        }
        // ---- the number of stuffed zeros before the group.  A group of ONE round: its count goes out here, the look-back follows at once.
        // A group of SEVERAL rounds (MULTI) COUNTS all its rounds first (pass 0: walk, window, census — nothing is stored), lets its count
        // out, looks back, and only then walks its rounds again to expand and store them (pass 1).  (Until the second session of round 6
        // such a group looked back in its first round and let its count out in its last: every group waited for the whole group before
        // it — noise at q >= 90, photographs at q = 100, gray noise at q = 80 took 24-46 ms per 4096x4096 file instead of 0.1 ms,
        // profiles/r06_long_groups_chain.txt.)
        if (MULTI && pass == 0) { // the count pass ends here
            if (wbase == 0 && tid == 0) s_first_ones = (x[0] >> 24) == (0xFFu >> sh8) ? 1u : 0u;
            ff_group += round_ff;
            if (tid == 0) s_carry = buf[wn - 1];
            __syncthreads();
            continue;
        }
        if (!MULTI) {
            if (tid == 0) store_relaxed(&desc2[g], kFlagAggregate | (uint64_t)round_ff);
            if (!second_look_back(round_ff)) return;
        if (!MULTI && seg_late && nsegs > 1) { // (workgroup-uniform) a segment's last group of one round: the segment's bytes are known — out
                                                // they go, and only then: where does the segment begin?
            if (tid == 0 && seg + 1 < nsegs) publish_aggregate(segdesc, seg, 0, (S >> 3) + nb_total + ff_before_groups + round_ff + rest.gap);
            if (seg > 0) {
                if (wave == 1) {
                    const uint64_t sb = look_back(segdesc, seg, 0, 0, a_state, host_abort, rest.spin_budget);
                    if (lane == 0) {
                        if (sb == kLookBackFailed) s_abort = 1;
                        s_segbase = sb;
                    }
                }
                __syncthreads();
                if (uni(s_abort)) { aborted = true; return; }
                seg_base = uni64(s_segbase);
            }
        }
        }
        // ---- expand + store.  Output offset of the round's first byte (owned byte 4 wbase of the group):
        const uint64_t dst_round = (uint64_t)rest.out_skew + seg_base + (S >> 3) + ff_before_groups + round_first + ff_group;
        const bool all_at_once = ((uint32_t)dst_round & 15u) + bytes_this + round_ff + 4u <= kStageCap;
        for (int turn = 0; turn < (all_at_once ? 1 : kGroupWaves); turn++) { // (wave by wave when a round's bytes + zeros do not fit the stage)
            const uint32_t before_turn = all_at_once ? 0u : (turn == 0 ? 0u : (turn == 1 ? ff_of_wave[0] : ff_of_wave[0] + ff_of_wave[1]));
            const uint64_t dst0 = dst_round + (all_at_once ? 0u : 2048u * (uint32_t)turn + before_turn);
            const uint32_t skew = (uint32_t)(dst0 & 15);
            const uint32_t turn_bytes = all_at_once ? bytes_this : (bytes_this > 2048u * (uint32_t)turn ? (bytes_this - 2048u * (uint32_t)turn < 2048u ? bytes_this - 2048u * (uint32_t)turn : 2048u) : 0u);
            const uint32_t turn_ff = all_at_once ? round_ff : (turn == 0 ? ff_of_wave[0] : (turn == 1 ? ff_of_wave[1] : ff_of_wave[2]));
            const uint32_t tile_out = turn_bytes + turn_ff;
            if (turn > 0) { // (the first turn's stage was zeroed on the way here)
                for (uint32_t i = 16u * tid; i < ((skew + tile_out + 15u) & ~15u); i += 16u * kGroup) *reinterpret_cast<v4u *>(stage + i) = v4u{0, 0, 0, 0};
                __syncthreads();
            }
            if (all_at_once || wave == turn) {
                const uint32_t at0 = skew + (all_at_once ? 2048u * (uint32_t)wave + wave_base_ff : 0u) + 4u * (uint32_t)lane;
#pragma unroll
                for (int k = 0; k < kRows; k++) {
                    if (4u * (wbase + 512u * (uint32_t)wave + 64u * k) >= limit) continue; // (wave-uniform)
                    const uint32_t first_byte = 4u * (wbase + jl0 + 64u * k);
                    const uint32_t m4 = (uint32_t)(flags >> (4 * k)) & 0xFu;
                    const uint32_t to = at0 + 256u * k + before[k];
                    // ONE condition per word, not per byte (32 exec masks a round were 64 spilled scalar registers): the word that
                    // holds the round's last byte writes up to three bytes behind it — into stage bytes nothing reads (the
                    // stores below stop at the round's end; all_at_once keeps 4 bytes of the stage free for them)
                    if (first_byte < limit) {
                        const uint32_t f0 = m4 & 1u, f1 = (m4 >> 1) & 1u, f2 = (m4 >> 2) & 1u;
                        const uint32_t a1 = to + 1u + f0, a2 = a1 + 1u + f1, a3 = a2 + 1u + f2;
                        stage[to] = (uint8_t)(x[k] >> 24);
                        stage[a1] = (uint8_t)(x[k] >> 16);
                        stage[a2] = (uint8_t)(x[k] >> 8);
                        stage[a3] = (uint8_t)x[k];
                    }
                    __builtin_amdgcn_sched_barrier(0); // (row by row: with all 32 addresses computed up front the byte stage spilled vector registers)
                }
            }
            __syncthreads();
            PIXO_STAMP(9);
            // out: leading bytes up to the first aligned 16 bytes, aligned 16-byte pieces, trailing bytes — never beyond out_cap
            const uint64_t base = dst0 - skew; // multiple of 16
            const uint32_t end = skew + tile_out;
            const uint32_t first_q = skew ? 16u : 0u, last_q = end & ~15u;
            if (first_q <= last_q) {
                for (uint32_t i = first_q + 16u * tid; i < last_q; i += 16u * kGroup)
                    if (base + i + 16 <= rest.out_cap) __builtin_nontemporal_store(*reinterpret_cast<const v4u *>(stage + i), reinterpret_cast<v4u *>(out + base + i));
                if (tid < 16) { // bytes [skew, min(16, end)) and [last_q, end)
                    const uint32_t i = skew + tid;
                    if (skew && i < 16 && i < end && base + i < rest.out_cap) out[base + i] = stage[i];
                    const uint32_t j = last_q + tid;
                    if (j < end && j >= first_q && base + j < rest.out_cap) out[base + j] = stage[j];
                }
            } else { // the whole turn lies inside one 16-byte piece
                if ((uint32_t)tid + skew < end && tid < 16 && base + skew + tid < rest.out_cap) out[base + skew + tid] = stage[skew + tid];
            }
            if (MULTI || !all_at_once) __syncthreads();
        }
        ff_group += round_ff;
        if (MULTI) {
            if (tid == 0) s_carry = buf[wn - 1];
            __syncthreads();
        }
    }
    }
    };
    if (park) rounds(std::true_type{}); else rounds(std::false_type{});
    PIXO_STAMP(10);
    if (aborted) return;
    if (last_group && tid == 0) { // the segment is complete: where the next one begins, where this one ends, the launch's totals
        const uint64_t packed = (S >> 3) + nb_total, stuffed = packed + ff_before_groups + ff_group, end = seg_base + stuffed;
        if (seg + 1 < nsegs) {
            store_relaxed(&segdesc[seg], kFlagPrefix | (end + rest.gap)); // (inclusive prefix: where the next segment begins)
            if (rest.rst) { // RSTn behind a restart interval while more MCUs follow (jpeg/mod.rs:1431-1445); never stuffed
                const uint64_t at = (uint64_t)rest.out_skew + end;
                if (at + 2 <= rest.out_cap) { out[at] = 0xFF; out[at + 1] = (uint8_t)(0xD0u + (srow & 7u)); }
            }
        }
        if (rest.host_segs) rest.host_segs[seg] = end;
        if (seg + 1 == nsegs) {
            const uint64_t bits = S + group_bits;
            a_state[1] = bits;
            if (rest.host_totals) { rest.host_totals[0] = bits; rest.host_totals[1] = end; rest.host_totals[2] = packed; }
        }
    }
#undef tid
#undef lane
#undef wave
}
// ---- statistics for optimised tables FROM PIXELS (second session of round 6) ------------------------------------------------------
// The reference's optimised-Huffman encode runs its pixel pipeline twice — once to count symbols, once to code
// (src/jpeg/mod.rs:659-860 count_block over every MCU, then encode_scan) — and so does the fused path now: this kernel is phases
// A and B of pixels_code_kernel followed by the flat walk as a COUNTER (block_count_flat: LDS counters per workgroup), the tuple is
// never written.  What crosses workgroups is one thing only — the DC predictor of a tile's first block of each component — and that
// needs no waiting: every workgroup leaves the DCs of its first and last block per component in HBM, does NOT count its first
// blocks' DC symbols, and the summing kernel counts those 3 x tiles symbols from the pairs (last of tile t - 1, first of tile t;
// 0 in front of a restart interval's first tile, src/jpeg/mod.rs:1441-1444).  Counters: one row of 16-bit words per tile (a tile
// holds at most 192 x 63 symbols), added up by pixels_count_sum_kernel into the [class][12 DC + 256 AC] layout of launch_scan_count.
struct LdsBumpPc { // (jpeg_scan_fused.hip's LdsBump)
    uint32_t *hist; // the class's kWalkClassWords counters
    uint32_t dummy; // index of this lane's dummy counter, relative to hist
    bool live;
    __device__ __forceinline__ void bump(uint32_t slot, bool on, uint32_t amount)
    {
        (void)__hip_atomic_fetch_add(&hist[on && live ? slot : dummy], amount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};
template <int MODE, int LOAD, bool PACKED>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(6, 6))) void pixels_count_kernel
(const uint8_t *a_px, uint32_t a_W, uint32_t a_H, const float *a_qt, uint32_t a_units_x, uint32_t a_units_y, uint16_t *a_slab, int16_t *a_edges,
 unsigned long long *a_hist, const PEarly early)
{
    typedef Geo<MODE> G;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kFusedLds];
    __shared__ int16_t s_dc[kGroup];
    static_assert((kWalkWords + kGroup) * 4 <= kFusedLds, "the counters fit the dead planar tile");
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    __builtin_amdgcn_s_setprio(1);
    const uint32_t tx = blockIdx.x, ty = blockIdx.y;
    TileCtx c;
    c.px = a_px; c.y = c.cb = c.cr = nullptr; c.qt = a_qt;
    c.W = a_W; c.H = a_H; c.units_x = a_units_x; c.units_y = a_units_y; c.fast = 1;
    c.px_end = a_px + early.px_bytes;
    if constexpr (MODE == MGRAY) {
        phase_a_gray_row<LOAD>(c, tx, ty, wave, lane, lds);
    } else {
        constexpr int base = G::items / kWaves, extra = G::items % kWaves;
        const int first = (extra && wave < extra) ? wave * (base + 1) : extra * (base + 1) + (wave - extra) * base;
        const LaneAddr la = lane_addr<MODE>(c, tx, ty, lane);
        auto run = [&](auto count_tag) __attribute__((always_inline)) {
            constexpr int COUNT = decltype(count_tag)::value;
            uint32_t r[COUNT * G::item_regs];
#pragma unroll
            for (int j = 0; j < COUNT; j++) producer_load_item<MODE, LOAD>(c, la, tx, ty, first + j, lane, &r[j * G::item_regs]);
#pragma unroll
            for (int j = 0; j < COUNT; j++) {
                producer_fix_item<MODE, LOAD>(c, tx, first + j, lane, &r[j * G::item_regs]);
                producer_color_item<MODE, true>(first + j, lane, &r[j * G::item_regs], lds);
            }
        };
        if (extra && wave < extra) run(std::integral_constant<int, base + 1>{}); else run(std::integral_constant<int, base>{});
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
    const uint32_t tiles_x = gridDim.x, g = ty * tiles_x + tx;
    if (g == 0) // (the sums start from zero: cheaper here than a memset launch — the summing kernel runs behind this one)
        for (int i = tid; i < kTableWords; i += kGroup) a_hist[i] = 0;
    const uint32_t u0 = tx * (uint32_t)tile_units<MODE>();
    const uint32_t nvalid = a_units_x - u0 < (uint32_t)tile_units<MODE>() ? a_units_x - u0 : (uint32_t)tile_units<MODE>();
    uint32_t m, comp, sidx;
    bool first_of_comp, last_block_of_mcu_comp;
    if (MODE == MGRAY) {
        m = (uint32_t)tid; comp = 0; sidx = m; first_of_comp = last_block_of_mcu_comp = true;
    } else if (MODE == M420) {
        if (wave < 2) { m = (uint32_t)wave * 16u + ((uint32_t)lane >> 2); comp = 0; sidx = 6u * m + ((uint32_t)lane & 3u); first_of_comp = (lane & 3) == 0; last_block_of_mcu_comp = (lane & 3) == 3; }
        else { m = (uint32_t)lane & 31u; comp = 1u + ((uint32_t)lane >> 5); sidx = 6u * m + 3u + comp; first_of_comp = last_block_of_mcu_comp = true; }
    } else {
        m = (uint32_t)lane; comp = (uint32_t)wave; sidx = 3u * m + comp; first_of_comp = last_block_of_mcu_comp = true;
    }
    const bool live = m < nvalid;
    const int dc = (int)(int16_t)(uint16_t)(qw[0] & 0xFFFFu);
    s_dc[sidx] = (int16_t)dc;
    const bool external = m == 0 && first_of_comp;
    // a_edges: [tile][component][0 = the tile's first block's DC, 1 = its last block's]
    if (external) a_edges[((size_t)g * 3 + comp) * 2] = (int16_t)dc;
    if (m + 1 == nvalid && last_block_of_mcu_comp) a_edges[((size_t)g * 3 + comp) * 2 + 1] = (int16_t)dc;
    __syncthreads(); // every wavefront has consumed its planar rows (the area becomes the counters), s_dc is complete
    uint32_t *lhist = reinterpret_cast<uint32_t *>(lds);
    for (int i = tid; i < kWalkWords + kGroup; i += kGroup) lhist[i] = 0;
    const uint32_t back = MODE == MGRAY ? 1u : (MODE == M420 ? (comp == 0 ? (first_of_comp ? 3u : 1u) : 6u) : 3u);
    const int prev_dc = external ? 0 : (int)s_dc[sidx - back];
    __syncthreads();
    const uint32_t cls = comp ? 1u : 0u;
    LdsBumpPc h{lhist + cls * kWalkClassWords, (uint32_t)(kWalkWords + tid) - cls * kWalkClassWords, live};
    block_count_flat(qw, prev_dc, h, !external);
    __syncthreads();
    for (int i = tid; i < kWalkWords; i += kGroup) a_slab[(size_t)g * kWalkWords + i] = (uint16_t)lhist[i];
}

// blockIdx.x: 64 columns of the slab; blockIdx.y < kPcSumRows: rows blockIdx.y, + kPcSumRows, ... in four phases;
// the workgroup (0, kPcSumRows): the DC symbols of the tiles' first blocks from the (last, first) pairs
constexpr uint32_t kPcSumRows = 32;
__global__ __launch_bounds__(256) void pixels_count_sum_kernel(const uint16_t *slab, const int16_t *edges, uint32_t tiles, uint32_t tiles_x, uint32_t seg_rows,
                                                               uint32_t comps, unsigned long long *hist)
{
    __shared__ uint32_t part[4][64];
    __shared__ uint32_t dcs[2][16];
    if (blockIdx.y == kPcSumRows) {
        if (blockIdx.x != 0) return;
        if (threadIdx.x < 32) dcs[threadIdx.x >> 4][threadIdx.x & 15] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < 3u * tiles; i += 256u) {
            const uint32_t t = i / 3u, comp = i - 3u * t;
            if (comp >= comps) continue; // (a gray tile has one component)
            const bool seg_first = t % (seg_rows * tiles_x) == 0; // (a restart interval's — or the image's — first tile: predictors 0)
            const int prev = seg_first ? 0 : (int)edges[((size_t)(t - 1) * 3 + comp) * 2 + 1];
            const int diff = (int)(int16_t)((int)edges[((size_t)t * 3 + comp) * 2] - prev);
            const uint32_t s = scan_sign_bits(diff + (diff >> 31)), m = s < 32u ? s : 32u;
            atomicAdd(&dcs[comp ? 1 : 0][m & 15u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const int sym = walk_slot_symbol((int)(threadIdx.x & 15));
            const uint32_t n = dcs[threadIdx.x >> 4][threadIdx.x & 15];
            if (n && sym >= 0) atomicAdd(&hist[(threadIdx.x >> 4) * kClassSyms + sym], (unsigned long long)n);
        }
        return;
    }
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), phase = threadIdx.x >> 6;
    uint32_t n = 0;
    if (col < kWalkWords)
        for (uint32_t r = blockIdx.y * 4u + (uint32_t)phase; r < tiles; r += 4u * kPcSumRows) n += slab[(size_t)r * kWalkWords + col];
    part[phase][threadIdx.x & 63] = n;
    __syncthreads();
    if (phase == 0 && col < kWalkWords) {
        const unsigned long long sum = (unsigned long long)part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        const int sym = walk_slot_symbol(col % kWalkClassWords);
        if (sum && sym >= 0) atomicAdd(&hist[(col / kWalkClassWords) * kClassSyms + sym], sum);
    }
}
} // namespace

size_t pixels_count_scratch_bytes(const PixelsCodePlan &p) { return (size_t)p.groups * kWalkWords * 2 + (size_t)p.groups * 3 * 2 * 2 + 16; }

hipError_t launch_pixels_count(const void *d_px, uint32_t W, uint32_t H, bool gray, bool s420, const PixelsCodePlan &p, const float *d_qt, void *d_scratch,
                               unsigned long long *d_hist, hipStream_t s)
{
    if (!pixels_code_supported(W, H, gray, s420, 1, 0) || p.images != 1 || p.groups > 0x7FFFFFFFull || p.tiles_y > 65535u) return hipErrorInvalidValue;
    PEarly early;
    const size_t row_bytes = (size_t)W * (gray ? 1 : 3);
    early.px_stride = row_bytes * H;
    early.px_bytes = early.px_stride;
    uint16_t *slab = static_cast<uint16_t *>(d_scratch);
    int16_t *edges = reinterpret_cast<int16_t *>(slab + (((size_t)p.groups * kWalkWords + 7) & ~(size_t)7));
    const bool aligned = reinterpret_cast<uintptr_t>(d_px) % 4 == 0 && row_bytes % 4 == 0;
    const dim3 grid(p.tiles_x, p.tiles_y, 1);
    const uint8_t *px = static_cast<const uint8_t *>(d_px);
    const bool packed = packed_launch(p.groups);
#define PIXO_LAUNCH_CNT2(MODE, LOAD, PK) hipLaunchKernelGGL((pixels_count_kernel<MODE, LOAD, PK>), grid, dim3(kThreads), 0, s, px, W, H, d_qt, p.units_x, p.units_y, slab, edges, d_hist, early)
#define PIXO_LAUNCH_CNT(MODE, LOAD) do { if (packed) PIXO_LAUNCH_CNT2(MODE, LOAD, true); else PIXO_LAUNCH_CNT2(MODE, LOAD, false); } while (0)
    if (gray) { if (aligned) PIXO_LAUNCH_CNT(MGRAY, L_ALIGNED); else PIXO_LAUNCH_CNT(MGRAY, L_FUNNEL); }
    else if (s420) { if (aligned) PIXO_LAUNCH_CNT(M420, L_ALIGNED); else PIXO_LAUNCH_CNT(M420, L_FUNNEL); }
    else { if (aligned) PIXO_LAUNCH_CNT(M444, L_ALIGNED); else PIXO_LAUNCH_CNT(M444, L_FUNNEL); }
#undef PIXO_LAUNCH_CNT
#undef PIXO_LAUNCH_CNT2
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pixels_count_sum_kernel, dim3((kWalkWords + 63) / 64, kPcSumRows + 1), dim3(256), 0, s, slab, edges, (uint32_t)p.groups, p.tiles_x, p.seg_rows, gray ? 1u : 3u, d_hist);
    return hipGetLastError();
}

#ifdef PIXO_TIMELINE
extern "C" __attribute__((visibility("default"))) int pixo_hip_debug_pixels_code_timeline(unsigned long long *out, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pc_timeline), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

PixelsCodePlan pixels_code_plan(uint32_t W, uint32_t H, bool s420, uint32_t images, uint32_t restart_mcus, bool gray)
{
    PixelsCodePlan p;
    const uint32_t unit = (s420 && !gray) ? 16u : 8u, per_tile = gray ? (uint32_t)kGrayTileBlocks : (s420 ? 32u : 64u);
    p.units_x = (W + unit - 1) / unit; p.units_y = (H + unit - 1) / unit;
    p.tiles_x = (p.units_x + per_tile - 1) / per_tile; p.tiles_y = p.units_y;
    p.images = images;
    p.seg_rows = p.tiles_y;
    if (restart_mcus && restart_mcus % p.units_x == 0 && restart_mcus / p.units_x < p.tiles_y) p.seg_rows = restart_mcus / p.units_x;
    p.segs_per_img = (p.tiles_y + p.seg_rows - 1) / p.seg_rows;
    p.seg_blocks64 = (uint32_t)(((uint64_t)p.seg_rows * p.tiles_x + 63) / 64);
    p.groups = (uint64_t)p.tiles_x * p.tiles_y * images;
    p.segments = (uint64_t)p.segs_per_img * images;
    // the block sums in copies (look_back_blocks): a launch of 256 groups or more keeps 16, each 4 KiB + 256 B behind the one before
    // (or the sums' own size rounded up to 256 B, + 256 B), so that the copies lie in different memory channels
    const size_t sums = (size_t)p.segments * p.seg_blocks64 + 1;
    const SupLayout sl = sup_layout(p.groups, sums);
    p.sup_copies = sl.copies;
    p.sup_stride = sl.stride;
    // abort flag, total bits; per group: bit-count descriptor, tail, three DC words, 0xFF-count descriptor; two rows of block sums
    // (in copies); per segment its byte count (+ 1)
    p.state_words = 2 + 6 * (size_t)p.groups + 2 * (size_t)p.sup_copies * p.sup_stride + (size_t)p.segments + 1;
    return p;
}

bool pixels_code_supported(uint32_t W, uint32_t H, bool gray, bool s420, uint32_t images, uint32_t restart_mcus)
{ // vector pixel loads need one whole 4-pixel group per row; a tile row per grid row; restart intervals: whole MCU rows of ONE image
    if (W < 4 || H < 1 || (H + 7) / 8 > 65535u || images < 1 || images > 65535u) return false;
    if (restart_mcus) {
        const uint32_t unit = (s420 && !gray) ? 16u : 8u, units_x = (W + unit - 1) / unit;
        if (images > 1 || restart_mcus % units_x != 0) return false;
    }
    return true;
}

hipError_t launch_pixels_code(const void *d_px, uint32_t W, uint32_t H, bool gray, bool s420, const PixelsCodePlan &p, uint32_t gap, bool rst_markers,
                              const float *d_qt, const uint32_t *d_tables, unsigned long long *d_state, bool state_is_zero,
                              unsigned long long *d_clear, size_t clear_words, uint8_t *d_out, uint64_t out_cap, unsigned long long *host_totals,
                              unsigned long long *host_segs, const int16_t seed_dc[3], bool pad_last, void *d_block_spill, hipStream_t s,
                              uint32_t spin_budget)
{
    if (!pixels_code_supported(W, H, gray, s420, p.images, 0) || clear_words > 0xFFFFFFFFull) return hipErrorInvalidValue;
    if (p.groups > 0x7FFFFFFFull || p.tiles_y > 65535u) return hipErrorInvalidValue;
    if (rst_markers && gap != 2) return hipErrorInvalidValue;
    if (host_totals) host_totals[3] = 0; // (the kernels' abort flag)
    if (!state_is_zero) {
        const hipError_t e = hipMemsetAsync(d_state, 0, p.state_words * 8, s);
        if (e != hipSuccess) return e;
    }
    PRest rest;
    PEarly early;
    const size_t row_bytes = (size_t)W * (gray ? 1 : 3);
    early.px_stride = row_bytes * H;
    early.px_bytes = early.px_stride * p.images;
    rest.clear = d_clear; rest.clear_words = d_clear ? (uint32_t)clear_words : 0u;
    rest.host_totals = host_totals; rest.host_segs = host_segs; rest.spin_budget = spin_budget;
    rest.tiles_x = p.tiles_x; rest.tiles_y = p.tiles_y; rest.groups = (uint32_t)p.groups;
    rest.seg_rows = p.seg_rows; rest.segs_per_img = p.segs_per_img; rest.seg_blocks64 = p.seg_blocks64;
    rest.sup_copies = p.sup_copies; rest.sup_stride = p.sup_stride;
    rest.gap = gap; rest.rst = rst_markers ? 1u : 0u;
    for (int i = 0; i < 3; i++) rest.seed_dc[i] = seed_dc ? seed_dc[i] : (int16_t)0;
    rest.pad_last = pad_last ? 1 : 0;
    // (d_out may start anywhere: the kernel gets the 16-byte boundary below it and the distance)
    rest.out_skew = (uint32_t)(reinterpret_cast<uintptr_t>(d_out) & 15);
    uint8_t *out = d_out - rest.out_skew;
    rest.out_cap = out_cap + rest.out_skew;
    rest.block_spill = static_cast<uint32_t *>(d_block_spill);
    const bool aligned = reinterpret_cast<uintptr_t>(d_px) % 4 == 0 && row_bytes % 4 == 0 && (p.images == 1 || early.px_stride % 4 == 0);
    const dim3 grid(p.tiles_x, p.tiles_y, p.images);
    const uint8_t *px = static_cast<const uint8_t *>(d_px);
    const bool packed = packed_launch(p.groups); // (scalar or packed DCT passes and quantiser: jpeg_kernels.hpp)
    const bool segs = p.segments > 1;
    const DispatchGate gate(s, p.groups); // (held until the kernel is enqueued)
    rest.gate_slots = gate.mark().slots; rest.gate_seq = gate.mark().seq;
#define PIXO_LAUNCH_PC2(MODE, LOAD, PK) do { if (segs) hipLaunchKernelGGL((pixels_code_kernel<MODE, LOAD, PK, true>), grid, dim3(kThreads), 0, s, px, W, H, d_qt, p.units_x, p.units_y, d_tables, d_state, out, early, rest); \
                                             else hipLaunchKernelGGL((pixels_code_kernel<MODE, LOAD, PK, false>), grid, dim3(kThreads), 0, s, px, W, H, d_qt, p.units_x, p.units_y, d_tables, d_state, out, early, rest); } while (0)
#define PIXO_LAUNCH_PC(MODE, LOAD) do { if (packed) PIXO_LAUNCH_PC2(MODE, LOAD, true); else PIXO_LAUNCH_PC2(MODE, LOAD, false); } while (0)
    if (gray) { if (aligned) PIXO_LAUNCH_PC(MGRAY, L_ALIGNED); else PIXO_LAUNCH_PC(MGRAY, L_FUNNEL); }
    else if (s420) { if (aligned) PIXO_LAUNCH_PC(M420, L_ALIGNED); else PIXO_LAUNCH_PC(M420, L_FUNNEL); }
    else { if (aligned) PIXO_LAUNCH_PC(M444, L_ALIGNED); else PIXO_LAUNCH_PC(M444, L_FUNNEL); }
#undef PIXO_LAUNCH_PC
#undef PIXO_LAUNCH_PC2
    return hipGetLastError();
}

} // namespace pixo_dev
