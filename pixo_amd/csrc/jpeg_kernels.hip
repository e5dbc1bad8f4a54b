// jpeg_kernels.hip — gfx950 kernels of the JPEG pixel pipeline and their launcher.
//
// The per-tile body lives in jpeg_tile.h (shared with the CPU emulation harness in
// tests/emu).  This file adds the __global__ wrapper, the tile -> image mapping and the
// host-side launch function.  Written for CDNA4 only: 64-lane wavefronts, 192-thread
// workgroups (3 waves), one 512-pixel-wide tile per workgroup, 2048 workgroups per 4096x4096
// 4:2:0 image (8 per CU, all resident at once).
#include <hip/hip_runtime.h>

#include <atomic>


#include "jpeg_kernels.hpp"
#include "jpeg_tile.h"

#pragma clang fp contract(off)

namespace pixo_dev {
using namespace pixo_tile;


// Kernel arguments.  The kernel takes the first 15 dwords as separate parameters, in this order, and is compiled with
// -mllvm -amdgpu-kernarg-preload-count=14 (Makefile): the command processor hands them to every wavefront in SCALAR
// REGISTERS at launch.  Read from the kernarg segment with s_load instead, they are the first thing every wavefront waits
// for — and the wavefronts that start a microsecond late queue behind the 50 MB of pixel loads the early ones have already
// issued: 0.25 us for the first wavefront, 1.5 us for the median, 6 us for the last (profiles/r03_timeline_c2_before.txt).
// Everything else (KRest) is not needed before the pixel loads are out and comes by s_load as before.
struct KRest {
    size_t px_bytes;  // all pixel bytes of the launch (the host emulation checks every vector load against them: jpeg_tile.h emu_check_load)
    size_t y_stride;  // i16 elements between consecutive images of a batch
    size_t c_stride;
    float *ry, *rcb, *rcr;   // RAW kernels only: unquantised DCT blocks (64 f32 each) instead of y/cb/cr
    uint32_t units_x, units_y, fast;
};
struct KArgs {
    const uint8_t *px;
    uint32_t W, H;
    size_t px_stride; // bytes between consecutive images of a batch
    int16_t *y, *cb, *cr;
    const float *qt;
    // ---- (14 dwords up to here) ----
    size_t px_bytes, y_stride, c_stride;
    float *ry, *rcb, *rcr;
    uint32_t units_x, units_y, fast;
    uint32_t tiles_x, tiles_y, batch; // host only
};

// Workgroup barrier that orders LDS only.  __syncthreads() also drains vmcnt, which would
// make every barrier wait for the next tile's prefetch loads and the previous tile's stores.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}


// A tile is (image, tile column, tile row): three SGPRs.  The full TileCtx is rebuilt from the
// kernel arguments where it is needed instead of being carried (three live copies of it cost
// ~30 SGPRs and pushed the kernel to the 102-SGPR limit).
struct TileId {
    uint32_t img, tx, ty;
};

// what phase A needs (pixels in); the coefficient arrays are filled in after the barrier (ctx_out): their address
// arithmetic has no business in front of the pixel loads
__device__ __forceinline__ TileCtx ctx_in(const KArgs &a, uint32_t img)
{
    TileCtx c;
    c.px = a.px + (size_t)img * a.px_stride;
    c.y = c.cb = c.cr = nullptr;
    c.qt = a.qt;
    c.W = a.W; c.H = a.H; c.units_x = a.units_x; c.units_y = a.units_y; c.fast = a.fast;
    c.px_end = a.px + a.px_bytes;
    return c;
}
__device__ __forceinline__ void ctx_out(TileCtx &c, const KArgs &a, uint32_t img)
{
    c.y = a.y + (size_t)img * a.y_stride;
    c.cb = a.cb ? a.cb + (size_t)img * a.c_stride : nullptr;
    c.cr = a.cr ? a.cr + (size_t)img * a.c_stride : nullptr;
}

// Phase A of one wavefront: COUNT items [first, first + COUNT) of the tile, HBM -> registers ->
// planar LDS.  All loads are issued before the first conversion (no branch near a load).
template <int MODE, int LOAD, int COUNT>
__device__ __forceinline__ void phase_a(const KArgs &a, const TileCtx &c, const TileId &id, int first, int lane, uint8_t *lds)
{
    typedef Geo<MODE> G;
    uint32_t r[COUNT * G::item_regs];
    LaneAddr la{};
    if (LOAD != L_BYTES) la = lane_addr<MODE>(c, id.tx, id.ty, lane); // (the byte gathers address every pixel by themselves)
#pragma unroll
    for (int j = 0; j < COUNT; j++) producer_load_item<MODE, LOAD>(c, la, id.tx, id.ty, first + j, lane, &r[j * G::item_regs]);
#pragma unroll
    for (int j = 0; j < COUNT; j++) {
        producer_fix_item<MODE, LOAD>(c, id.tx, first + j, lane, &r[j * G::item_regs]);
        producer_color_item<MODE, LOAD != L_BYTES>(first + j, lane, &r[j * G::item_regs], lds);
    }
}

// One tile per workgroup of THREE wavefronts.  Phase A: the wavefronts share the tile's items
// 6/5/5 (16/16/16 for grey), load them and colour-convert them into planar LDS; one LDS barrier;
// phase B: each wavefront transforms, quantises and stores 64 blocks (one per lane), staging the
// quantised rows in the LDS area its own planar samples came from — it has consumed them all by
// then — so phase B needs no barrier and the workgroup only ~17 KiB of LDS.
// Three waves, not four: the tile has exactly 3 x 64 blocks, so a fourth wavefront would idle
// through phase B (70 % of the work) while holding a wave slot; with 3-wave workgroups the 24
// wave slots of a CU (79 VGPRs -> 6 per SIMD) hold all 8 tiles a CU gets of a 4096x4096 image at
// once: every load is issued in the first microsecond and the VALU stays saturated to the end
// (a single wavefront can issue only every ~4.3 cycles; two per SIMD are needed to fill it).
// Measured alternatives (4-wave workgroups; persistent loop with register prefetch;
// role-specialised producer/consumer wavefronts with double-buffered LDS): DESIGN.md.
// 6 waves per SIMD = 24 per CU = the 8 three-wave workgroups a CU gets of a 4096x4096 image: the
// compiler keeps every variant at 80 VGPRs without spilling when told to.
#define PIXO_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(6, 6)))
// RAW kernels: the lane's transformed block as 64 floats (natural order, the reference's dct_2d
// output: 4:2:0 chroma scaled back by the exact factor 1/4) for the trellis quantiser.
template <int MODE>
__device__ __forceinline__ void store_raw_block(const KArgs &a, const TileCtx &c, const TileId &id, int wave, int lane, const float *v)
{
    typedef Geo<MODE> G;
    const uint32_t u0 = id.tx * G::units_x;
    const uint32_t nvalid = c.units_x - u0 < (uint32_t)G::units_x ? c.units_x - u0 : G::units_x;
    float *dst = nullptr;
    float scale = 1.0f;
    if (MODE == M420) {
        const size_t mcu0 = (size_t)id.ty * c.units_x + u0;
        if (wave < 2) {
            if ((uint32_t)(wave * 16 + (lane >> 2)) < nvalid) dst = a.ry + (size_t)id.img * a.y_stride + (mcu0 * 4 + wave * 64 + lane) * 64;
        } else {
            const int m = lane & 31;
            if ((uint32_t)m < nvalid) dst = (lane < 32 ? a.rcb : a.rcr) + (size_t)id.img * a.c_stride + (mcu0 + m) * 64;
            scale = kChromaSumScale;
        }
    } else if (MODE == M444) {
        const size_t blk = (size_t)id.ty * c.units_x + u0 + lane;
        if ((uint32_t)lane < nvalid) {
            if (wave == 0) dst = a.ry + (size_t)id.img * a.y_stride + blk * 64;
            else dst = (wave == 1 ? a.rcb : a.rcr) + (size_t)id.img * a.c_stride + blk * 64;
        }
    } else {
        const uint32_t brow = id.ty * 3 + wave;
        if ((uint32_t)lane < nvalid && brow < c.units_y) dst = a.ry + (size_t)id.img * a.y_stride + ((size_t)brow * c.units_x + u0 + lane) * 64;
    }
    if (!dst) return;
    // The trellis kernel's layout: the three planes are one run of blocks (launch_jpeg_coeffs checks it), block B's
    // coefficient i lies at [B / 64][i][B % 64] — the 64 blocks one wavefront of that kernel searches side by side, so
    // that each of its steps is one coalesced load instead of a staging area in LDS.  Here: consecutive lanes hold
    // consecutive blocks, every one of the 64 stores below is a run of dwords.
    const size_t B = (size_t)(dst - a.ry) >> 6;
    float *col = a.ry + ((B >> 6) << 12) + (B & 63);
#pragma unroll
    for (int i = 0; i < 64; i++) col[i * 64] = v[i] * scale;
}

template <int MODE, int LOAD, bool RAW, bool PACKED>
__global__ __launch_bounds__(kThreads) PIXO_WAVES_ATTR void jpeg_coeffs_kernel(const uint8_t *a_px, uint32_t a_W, uint32_t a_H, size_t a_px_stride, uint32_t a_grid,
                                                                              uint32_t a_unused, int16_t *a_y, int16_t *a_cb, int16_t *a_cr, const float *a_qt, const KRest rest)
{
    typedef Geo<MODE> G;
    __shared__ __attribute__((aligned(16))) uint8_t lds[G::planar];
    KArgs a;
    a.px = a_px; a.W = a_W; a.H = a_H; a.px_stride = a_px_stride; a.y = a_y; a.cb = a_cb; a.cr = a_cr; a.qt = a_qt;
    a.px_bytes = rest.px_bytes; a.y_stride = rest.y_stride; a.c_stride = rest.c_stride; a.ry = rest.ry; a.rcb = rest.rcb; a.rcr = rest.rcr;
    a.units_x = rest.units_x; a.units_y = rest.units_y; a.fast = rest.fast;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // Phase A runs at raised wave priority: the hardware otherwise issues oldest-first, the colour
    // conversion of the younger workgroups waits behind the older ones' phase B, and few wavefronts
    // are in phase B at any time (two per SIMD are needed to fill the VALU).  +14 % for one 4096x4096
    // image (all workgroups resident at once), +10 % for 4:4:4 and for the 64-image batch (measured at
    // steady clocks, profiles/r01_ablation_steady_clocks.txt).
    __builtin_amdgcn_s_setprio(1);
    // The chip holds 2048 of these workgroups at once (8 per CU): a 4096x4096 launch is ONE generation whose loads all
    // come first and whose stores all come last.  The second half of that generation (dispatch order 1024..2047) starts
    // one s_sleep (8128 clocks, ~3.4 us) late, so that its loads meet the first half's arithmetic and stores: 19.1 ->
    // 18.35 us on one box of the pool, nothing gained or lost (17.4) on a faster one, two sleeps or other halves worse
    // (profiles/r03_stagger_and_occupancy_ab.txt); the 4:4:4 kernel, two generations per 4096x4096 image: 33.8 -> 30.9 us
    // (profiles/r03_stagger_444.txt; round 6 tried every odd thousand, halves of 512, graded quarters and two sleeps for 4:4:4 again:
    // all 1.5-3.5 us slower than this, none at all 0.45 us slower — profiles/r06_ab_444.txt).  Later generations are not touched
    // (every odd thousand late: the batch loses 6 %).
    // With the grid's shape pre-loaded (below) the gain is 19.9-20.6 -> 18.3 us on a medium box and 21.8 -> 19.1 us on a slow
    // one (profiles/r03_stagger_length_ab.txt); on the fastest kind of box it was within +-1 %.
    // (the grid's shape comes as a preloaded argument — tiles_x | tiles_y << 8 | "2048 workgroups or more" << 31: gridDim
    // would be a scalar load from the hidden arguments and a wait for it in front of every wavefront's first instruction)
    if (a_grid >> 31) {
        const uint32_t gx = a_grid & 0xFFu, gy = (a_grid >> 8) & 0xFFFFu;
        const uint32_t lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        if ((lin >> 10) == 1u) __builtin_amdgcn_s_sleep(127); // (a full first generation only; 112..160 units are a plateau, 96 or 200 lose most of it;
                                                                 //  quarters or thirds with graded delays are no better)
    }
    // (a 3-D grid: no division on the way to the first load.  Rows that are not dword aligned read 9.6 % more from HBM — the line at a
    // tile's edge is shared with the row's next tile, which round-robin dispatch puts behind another XCD's L2; keeping a row's tiles on
    // one XCD instead changed nothing in time: profiles/r06_unaligned_rows_per_xcd.txt)
    const TileId id{blockIdx.z, blockIdx.x, blockIdx.y};
    TileCtx c = ctx_in(a, id.img);
    constexpr int base = G::items / kWaves, extra = G::items % kWaves;
    const int first = (extra && wave < extra) ? wave * (base + 1) : extra * (base + 1) + (wave - extra) * base;
    if (extra && wave < extra) phase_a<MODE, LOAD, base + 1>(a, c, id, first, lane, lds);
    else phase_a<MODE, LOAD, base>(a, c, id, first, lane, lds);
    lds_barrier();
    __builtin_amdgcn_s_setprio(0);
    {
        uint32_t img = id.img;
        asm volatile("" : "+s"(img)); // (keeps the output pointers' arithmetic behind the barrier)
        ctx_out(c, a, img);
    }
    float v[64];
    consumer_rows<MODE, PACKED>(wave, lane, lds, v);
    consumer_cols<PACKED>(v);
    if (RAW) { // hand the transformed blocks to the trellis quantiser instead of quantising here
        store_raw_block<MODE>(a, c, id, wave, lane, v);
        return;
    }
    uint8_t *stage = lds + stage_offset<MODE>(wave); // inside this wavefront's own planar area
    uint32_t qw[32];
    consumer_quant<MODE, PACKED>(wave, lane, a.qt, v, qw);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        consumer_stage_blocks(lane, h, qw, stage);
        consumer_stage_sync();
        consumer_store_blocks<MODE>(c, id.tx, id.ty, wave, lane, h, stage);
        consumer_stage_sync();
    }
}

// (a debug switch of tools and tests — pixo_hip_debug_configure may flip it while other threads launch: a relaxed atomic)
static std::atomic<int> g_coef_form{0};
void set_coef_form(int form) { g_coef_form.store(form, std::memory_order_relaxed); }
bool packed_launch(uint64_t workgroups)
{
    const int form = g_coef_form.load(std::memory_order_relaxed);
    return form == 1 ? false : (form == 2 ? true : workgroups > 2048);
}

// The late start of dispatch numbers 1024..2047 assumes that 2048 workgroups are ONE resident generation: the whole MI355X
// (256 CUs x 8 workgroups).  A partitioned device (CPX / a compute partition with fewer CUs) holds fewer: no stagger there.
bool whole_chip_device()
{
    static std::atomic<int> cus[64]; // 0: not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = -1;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n == 256;
}

template <int MODE, int LOAD> static hipError_t launch_mode(KArgs &a, hipStream_t s)
{
    const bool raw = a.ry != nullptr;
    a.tiles_x = (a.units_x + Geo<MODE>::units_x - 1) / Geo<MODE>::units_x;
    a.tiles_y = (a.units_y * (MODE == M420 ? 16u : 8u) + Geo<MODE>::tile_h - 1) / Geo<MODE>::tile_h;
    if (a.tiles_y > 65535u || a.batch > 65535u) return hipErrorInvalidValue; // (grid y and z; heights and batches are at most 65535)
    const dim3 grid(a.tiles_x, a.tiles_y, a.batch); // workgroups are numbered x fastest: consecutive tiles of a row on consecutive XCDs
    KRest rest;
    rest.px_bytes = a.px_bytes; rest.y_stride = a.y_stride; rest.c_stride = a.c_stride; rest.ry = a.ry; rest.rcb = a.rcb; rest.rcr = a.rcr;
    rest.units_x = a.units_x; rest.units_y = a.units_y; rest.fast = a.fast;
    // tiles_x <= 128 (65535 / 512), tiles_y <= 8192 (65535 / 8)
    // (launches of 1280-1792 workgroups neither gain nor lose with it: tools/shape_timing.py, profiles/r03_stagger_length_ab.txt)
    const uint32_t a_grid = grid.x | (grid.y << 8) | (((uint64_t)grid.x * grid.y * grid.z >= 2048u && whole_chip_device()) ? 0x80000000u : 0u);
    // one generation of workgroups (at most 2048: all resident at once): the scalar forms of the DCT passes and the quantiser;
    // more: the packed forms (jpeg_tile.h, block_rows)
    const bool packed = packed_launch((uint64_t)grid.x * grid.y * grid.z);
#define PIXO_LAUNCH_C(RAW, PACKED) hipLaunchKernelGGL((jpeg_coeffs_kernel<MODE, LOAD, RAW, PACKED>), grid, dim3(kThreads), 0, s, a.px, a.W, a.H, a.px_stride, a_grid, 0u, a.y, a.cb, a.cr, a.qt, rest)
    if (raw) { if (packed) PIXO_LAUNCH_C(true, true); else PIXO_LAUNCH_C(true, false); }
    else { if (packed) PIXO_LAUNCH_C(false, true); else PIXO_LAUNCH_C(false, false); }
#undef PIXO_LAUNCH_C
    return hipGetLastError();
}

template <int MODE> static hipError_t launch_load(int load, KArgs &a, hipStream_t s)
{
    if (load == L_ALIGNED) return launch_mode<MODE, L_ALIGNED>(a, s);
    if (load == L_FUNNEL) return launch_mode<MODE, L_FUNNEL>(a, s);
    return launch_mode<MODE, L_BYTES>(a, s);
}

// The quantised DC of the last block of every plane — what the next band of a multi-GPU image predicts from — stored
// straight into pinned host memory by one thread: one launch instead of three 2-byte copies.
__global__ void last_dcs_kernel(const int16_t *y, size_t y_blocks, const int16_t *cb, const int16_t *cr, size_t c_blocks, int16_t *out)
{
    out[0] = y[(y_blocks - 1) * 64];
    out[1] = c_blocks ? cb[(c_blocks - 1) * 64] : (int16_t)0;
    out[2] = c_blocks ? cr[(c_blocks - 1) * 64] : (int16_t)0;
}
hipError_t launch_last_dcs(const void *d_y, size_t y_blocks, const void *d_cb, const void *d_cr, size_t c_blocks, int16_t *host_out, hipStream_t stream)
{
    hipLaunchKernelGGL(last_dcs_kernel, dim3(1), dim3(1), 0, stream, static_cast<const int16_t *>(d_y), y_blocks, static_cast<const int16_t *>(d_cb),
                       static_cast<const int16_t *>(d_cr), c_blocks, host_out);
    return hipGetLastError();
}

hipError_t launch_jpeg_coeffs(const void *d_px, uint32_t W, uint32_t H, bool gray, bool s420,
                              uint32_t batch, void *d_y, void *d_cb, void *d_cr,
                              const float *d_qt, hipStream_t stream, bool raw_f32)
{
    KArgs a;
    a.ry = a.rcb = a.rcr = nullptr;
    if (raw_f32) { // the same three planes back to back, 64 f32 per block, in the trellis kernel's layout (store_raw_block)
        a.ry = static_cast<float *>(d_y); a.rcb = static_cast<float *>(d_cb); a.rcr = static_cast<float *>(d_cr);
        if (batch != 1) return hipErrorInvalidValue;
    }
    a.px = static_cast<const uint8_t *>(d_px);
    a.y = static_cast<int16_t *>(d_y);
    a.cb = static_cast<int16_t *>(d_cb);
    a.cr = static_cast<int16_t *>(d_cr);
    a.qt = d_qt;
    a.W = W; a.H = H; a.batch = batch;
    const uint32_t unit = (!gray && s420) ? 16 : 8;
    a.units_x = (W + unit - 1) / unit;
    a.units_y = (H + unit - 1) / unit;
    const uint32_t bpp = gray ? 1 : 3;
    const size_t row_bytes = static_cast<size_t>(W) * bpp;
    a.px_stride = row_bytes * H;
    a.px_bytes = a.px_stride * batch;
    // one 12-byte load per lane and row needs 4-byte aligned rows in every image of the batch; other
    // images read aligned dwords and shift (L_FUNNEL); both need one whole 4-pixel group per row
    const bool aligned = (reinterpret_cast<uintptr_t>(d_px) % 4 == 0) && (row_bytes % 4 == 0) &&
                         (batch == 1 || a.px_stride % 4 == 0);
    const int load = W < 4 ? L_BYTES : (aligned ? L_ALIGNED : L_FUNNEL);
    a.fast = load != L_BYTES;
    const size_t units = static_cast<size_t>(a.units_x) * a.units_y;
    a.y_stride = (unit == 16 ? 4 * units : units) * 64;
    a.c_stride = units * 64;
    if (gray) return launch_load<MGRAY>(load, a, stream);
    if (s420) return launch_load<M420>(load, a, stream);
    return launch_load<M444>(load, a, stream);
}

} // namespace pixo_dev

