// jpeg_tile.h — the per-tile body of the fused JPEG coefficient kernel for gfx950.
//
// One 256-thread workgroup turns one tile of pixels into quantised DCT blocks:
//
//   phase A  (all 256 lanes, pixel-parallel)   global RGB8 -> registers (coalesced
//            12 B/lane = 4 px), integer BT.601 colour conversion with packed-u16
//            VALU ops (2 px per instruction), 2x2 chroma box sums, planar u8/u16
//            samples into LDS.
//   phase B  (one lane per 8x8 block)          LDS -> 64 f32 registers, level shift,
//            f32 AAN DCT rows then columns entirely in registers (no transposes),
//            quantise (reciprocal fast path proven equal to the IEEE divide, exact
//            divide fallback), pack to i16, swizzled 16-B chunks into an LDS stage.
//   phase C  (all lanes)                       LDS stage -> global, 16 B per lane,
//            fully coalesced, in the reference's YCbCrCoefficients layout.
//
// Reference semantics reproduced bit-for-bit (leerob/pixo v0.4.1):
//   colour           src/color.rs:60-77           (integer, 2^8-scaled, clamp)
//   extract + box    src/jpeg/mod.rs:1565-1656    (edge replicate; chroma = f32 mean
//                                                  of the 4 already-rounded u8 values)
//   DCT              src/jpeg/dct.rs:614-700      (f32 AAN, scale inside each pass)
//   quantise         src/jpeg/quantize.rs:99-105  ((x / q).round() as i16)
//
// The same source is compiled (a) by hipcc for the device and (b) by clang++ with
// -DPIXO_EMU for tests/emu, where a host harness runs the phases lane by lane; the
// -m "not gpu" test-suite therefore executes this very code against the oracle.
//
// Compile with -ffp-contract=off: every f32 operation of the DCT must round once,
// exactly as rustc emits it (no FMA).  The only fused operation below is an explicit
// fmaf in OUR safety test, which is not part of the reference arithmetic.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(PIXO_EMU)
#include <math.h>
#include <string.h>
#define PIXO_DEV static inline
#define PIXO_SCHED_FENCE() ((void)0)
#define PIXO_PIN(x) ((void)0)
#else
#define PIXO_DEV __device__ __forceinline__
// Stops the machine scheduler from interleaving independent 1-D transforms: left alone it
// sinks the whole column pass into the quantiser rows, keeping ~16 temporaries of all 8
// columns alive (150 VGPRs, 3 waves/SIMD).  Fenced, a block needs 64 + ~20 registers.
#define PIXO_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// Materialises a value here: stops LLVM from sinking the column pass into the quantiser's
// basic blocks (which kept every column's butterflies alive across them).
#define PIXO_PIN(x) asm volatile("" : "+v"(x))
#endif

#pragma clang fp contract(off)

namespace pixo_tile {

enum Mode { M420 = 0, M444 = 1, MGRAY = 2 };

constexpr int kThreads = 256;
constexpr int kTileW = 512;        // pixels per tile row
constexpr int kPitch = kTileW + 16; // LDS bytes per planar row: 8*kPitch % 256 == 128
                                    // puts the bottom Y blocks of an MCU on the other
                                    // half of the 64 banks (ds_read_b64 conflict-free)

// Quantiser table block for one quality, resident in HBM, read with scalar loads:
//   [0,64)    1/q luminance   [64,128)   1/q chrominance   (f32, correctly rounded)
//   [128,192) q   luminance   [192,256)  q   chrominance   (f32, exact integers 1..255)
constexpr int kQtFloats = 256;

template <int MODE> struct Geo;
template <> struct Geo<M420> {
    static constexpr int tile_h = 16, units_x = 32 /* MCUs */, blocks = 192;
    static constexpr int y_off = 0, cb_off = 16 * kPitch, cr_off = cb_off + 4096;
    static constexpr int planar_bytes = cr_off + 4096;
    static constexpr int stage_bytes = blocks * 128;
};
template <> struct Geo<M444> {
    static constexpr int tile_h = 8, units_x = 64 /* blocks */, blocks = 192;
    static constexpr int y_off = 0, cb_off = 8 * kPitch, cr_off = 16 * kPitch;
    static constexpr int planar_bytes = 24 * kPitch;
    static constexpr int stage_bytes = blocks * 128;
};
template <> struct Geo<MGRAY> {
    static constexpr int tile_h = 32, units_x = 64 /* blocks */, blocks = 256;
    static constexpr int y_off = 0, cb_off = 0, cr_off = 0;
    static constexpr int planar_bytes = 32 * kPitch;
    static constexpr int stage_bytes = blocks * 128;
};
template <int MODE> constexpr int lds_bytes()
{
    return Geo<MODE>::planar_bytes > Geo<MODE>::stage_bytes ? Geo<MODE>::planar_bytes
                                                             : Geo<MODE>::stage_bytes;
}

// Per-image launch context (uniform across the workgroup).
struct TileCtx {
    const uint8_t *px;   // this image's pixels, tightly packed rows
    int16_t *y, *cb, *cr; // this image's coefficient arrays
    const float *qt;     // kQtFloats floats for the requested quality
    uint32_t W, H;       // pixels
    uint32_t units_x;    // MCUs per row (4:2:0) or 8x8 blocks per row (4:4:4, gray)
    uint32_t units_y;    // MCU rows / block rows
    uint32_t fast;       // rows are 4-byte aligned: (px % 4 == 0) && (W*bpp % 4 == 0)
};

// ---------------------------------------------------------------------------------
// small intrinsic wrappers (device instruction / host emulation)
// ---------------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

PIXO_DEV uint32_t bits(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
PIXO_DEV u16x2 pk(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
PIXO_DEV u16x2 splat(unsigned v) { return u16x2{(unsigned short)v, (unsigned short)v}; }

// v_perm_b32: result byte i = byte sel[i] of the 64-bit value {hi,lo}; 0x0C -> 0x00.
PIXO_DEV uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
#if defined(PIXO_EMU)
    uint64_t both = ((uint64_t)hi << 32) | lo;
    uint32_t out = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t s = (sel >> (8 * i)) & 0xFF;
        uint32_t b = s <= 7 ? (uint32_t)((both >> (8 * s)) & 0xFF) : (s == 0x0C ? 0u : 0xFFu);
        out |= b << (8 * i);
    }
    return out;
#else
    return __builtin_amdgcn_perm(hi, lo, sel);
#endif
}

// two f32 (integer-valued) -> packed i16 pair, saturating like Rust's `as i16`
PIXO_DEV uint32_t pack_i16(float a, float b)
{
#if defined(PIXO_EMU)
    int ia = (int)a, ib = (int)b;
    ia = ia > 32767 ? 32767 : (ia < -32768 ? -32768 : ia);
    ib = ib > 32767 ? 32767 : (ib < -32768 ? -32768 : ib);
    return (uint32_t)(uint16_t)(int16_t)ia | ((uint32_t)(uint16_t)(int16_t)ib << 16);
#else
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    s16x2 p = __builtin_amdgcn_cvt_pk_i16((int)a, (int)b);
    return __builtin_bit_cast(uint32_t, p);
#endif
}

PIXO_DEV int uniform_i32(int v)
{
#if defined(PIXO_EMU)
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}

struct u32x2 { uint32_t x, y; };
struct alignas(16) u32x4 { uint32_t x, y, z, w; };

// ---------------------------------------------------------------------------------
// phase A: colour conversion of 4 horizontally adjacent pixels held as 3 dwords
//   d0 = R0 G0 B0 R1   d1 = G1 B1 R2 G2   d2 = B2 R3 G3 B3   (little-endian bytes)
// ---------------------------------------------------------------------------------
struct Row4 {
    uint32_t y4;         // Y0..Y3 as bytes
    u16x2 cb01, cb23;    // min(X'>>8, 254) per pixel  (= Cb - 1, see below)
    u16x2 cr01, cr23;
};

// color.rs:60-77 restated for packed u16 lanes.
//   Y  = (77R + 150G + 29B + 128) >> 8            max 65408: fits u16, clamp is a no-op.
//   Cb = ((-43R - 85G + 128B + 128) >> 8) + 128.  With X' = 128B + 32640 - 43R - 85G
//        (always in [0, 65280], so u16 arithmetic never wraps in the final value),
//        Cb = (X' >> 8) + 1, and the reference's clamp to 255 is min(X' >> 8, 254) + 1.
//   Cr = same with X' = 128R + 32640 - 107G - 21B.
// The "+1" is folded into the level shift of phase B (x - 127 instead of x - 128),
// which is exact.  Arithmetic >> on negative i32 in the reference equals the floor
// that the biased unsigned shift computes.
PIXO_DEV Row4 color_row4(uint32_t d0, uint32_t d1, uint32_t d2)
{
    u16x2 r01 = pk(perm(d0, d0, 0x0C030C00u));
    u16x2 g01 = pk(perm(d1, d0, 0x0C040C01u));
    u16x2 b01 = pk(perm(d1, d0, 0x0C050C02u));
    u16x2 r23 = pk(perm(d2, d1, 0x0C050C02u));
    u16x2 g23 = pk(perm(d2, d1, 0x0C060C03u));
    u16x2 b23 = pk(perm(d2, d2, 0x0C030C00u));

    u16x2 y01 = r01 * splat(77) + (g01 * splat(150) + (b01 * splat(29) + splat(128)));
    u16x2 y23 = r23 * splat(77) + (g23 * splat(150) + (b23 * splat(29) + splat(128)));

    const u16x2 kM43 = splat(65536 - 43), kM85 = splat(65536 - 85);
    const u16x2 kM107 = splat(65536 - 107), kM21 = splat(65536 - 21);
    u16x2 cb01 = r01 * kM43 + (g01 * kM85 + (b01 * splat(128) + splat(32640)));
    u16x2 cb23 = r23 * kM43 + (g23 * kM85 + (b23 * splat(128) + splat(32640)));
    u16x2 cr01 = g01 * kM107 + (b01 * kM21 + (r01 * splat(128) + splat(32640)));
    u16x2 cr23 = g23 * kM107 + (b23 * kM21 + (r23 * splat(128) + splat(32640)));

    Row4 o;
    o.y4 = perm(bits(y23), bits(y01), 0x07050301u); // high byte of each u16 lane
    const u16x2 k254 = splat(254);
    o.cb01 = __builtin_elementwise_min(cb01 >> 8, k254);
    o.cb23 = __builtin_elementwise_min(cb23 >> 8, k254);
    o.cr01 = __builtin_elementwise_min(cr01 >> 8, k254);
    o.cr23 = __builtin_elementwise_min(cr23 >> 8, k254);
    return o;
}

// Edge path: 4 pixels with the reference's clamp-replicate addressing
// (x = min(x, W-1), y = min(y, H-1); jpeg/mod.rs:1578-1579,1626-1627), byte loads.
struct u32x3 { uint32_t a, b, c; };

PIXO_DEV u32x3 gather_row4_rgb(const uint8_t *px, uint32_t W, uint32_t H, uint32_t x0, uint32_t y)
{
    const uint32_t yc = y < H ? y : H - 1;
    const uint8_t *row = px + (size_t)yc * W * 3;
    uint32_t v[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t x = x0 + i < W ? x0 + i : W - 1;
        v[3 * i] = row[x * 3]; v[3 * i + 1] = row[x * 3 + 1]; v[3 * i + 2] = row[x * 3 + 2];
    }
    u32x3 d;
    d.a = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
    d.b = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
    d.c = v[8] | (v[9] << 8) | (v[10] << 16) | (v[11] << 24);
    return d;
}

PIXO_DEV uint32_t gather_row4_gray(const uint8_t *px, uint32_t W, uint32_t H, uint32_t x0,
                                   uint32_t y)
{
    const uint32_t yc = y < H ? y : H - 1;
    const uint8_t *row = px + (size_t)yc * W;
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t x = x0 + i < W ? x0 + i : W - 1;
        d |= (uint32_t)row[x] << (8 * i);
    }
    return d;
}

// INTERIOR: the whole tile lies inside the image and rows are dword aligned, so every
// lane takes the 12-byte vector load with no per-item test.
template <bool INTERIOR>
PIXO_DEV void load_row4_rgb(const TileCtx &c, uint32_t x0, uint32_t y, uint32_t *d)
{
    if (INTERIOR || (c.fast && x0 + 4 <= c.W && y < c.H)) {
        const uint32_t *q = (const uint32_t *)(c.px + ((size_t)y * c.W + x0) * 3);
        d[0] = q[0]; d[1] = q[1]; d[2] = q[2];
    } else {
        u32x3 t = gather_row4_rgb(c.px, c.W, c.H, x0, y);
        d[0] = t.a; d[1] = t.b; d[2] = t.c;
    }
}

// Per-lane registers that live across the workgroup barriers.
template <int MODE> struct Lane {
    uint32_t in[MODE == M420 ? 24 : (MODE == M444 ? 12 : 16)];
    float v[64];
};

// ---- phase A.1: global loads (all issued before any use) ---------------------------
template <int MODE> PIXO_DEV bool tile_is_interior(const TileCtx &c, uint32_t tile_x, uint32_t tile_y)
{
    return c.fast && (tile_x + 1) * kTileW <= c.W && (tile_y + 1) * Geo<MODE>::tile_h <= c.H;
}

template <int MODE, bool INTERIOR>
PIXO_DEV void phase_load(const TileCtx &c, uint32_t tile_x, uint32_t tile_y, int tid, Lane<MODE> &L)
{
    const uint32_t tx0 = tile_x * kTileW, ty0 = tile_y * Geo<MODE>::tile_h;
    if (MODE == M420) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int item = k * kThreads + tid, g = item & 127, p = item >> 7;
            load_row4_rgb<INTERIOR>(c, tx0 + 4 * g, ty0 + 2 * p, &L.in[k * 6]);
            load_row4_rgb<INTERIOR>(c, tx0 + 4 * g, ty0 + 2 * p + 1, &L.in[k * 6 + 3]);
        }
    } else if (MODE == M444) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int item = k * kThreads + tid, g = item & 127, r = item >> 7;
            load_row4_rgb<INTERIOR>(c, tx0 + 4 * g, ty0 + r, &L.in[k * 3]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            int item = k * kThreads + tid, g = item & 127, r = item >> 7;
            uint32_t x0 = tx0 + 4 * g, y = ty0 + r;
            if (INTERIOR || (c.fast && x0 + 4 <= c.W && y < c.H))
                L.in[k] = *(const uint32_t *)(c.px + (size_t)y * c.W + x0);
            else
                L.in[k] = gather_row4_gray(c.px, c.W, c.H, x0, y);
        }
    }
}

// ---- phase A.2: colour + subsample -> planar LDS -----------------------------------
template <int MODE> PIXO_DEV void phase_color(int tid, const Lane<MODE> &L, uint8_t *lds)
{
    typedef Geo<MODE> G;
    if (MODE == M420) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int item = k * kThreads + tid, g = item & 127, p = item >> 7;
            Row4 a = color_row4(L.in[k * 6], L.in[k * 6 + 1], L.in[k * 6 + 2]);
            Row4 b = color_row4(L.in[k * 6 + 3], L.in[k * 6 + 4], L.in[k * 6 + 5]);
            *(uint32_t *)(lds + G::y_off + (2 * p) * kPitch + 4 * g) = a.y4;
            *(uint32_t *)(lds + G::y_off + (2 * p + 1) * kPitch + 4 * g) = b.y4;
            // 2x2 box sums (jpeg/mod.rs:1641-1646): vertical then horizontal, u16 exact
            uint32_t cb01 = bits(a.cb01 + b.cb01), cb23 = bits(a.cb23 + b.cb23);
            uint32_t cr01 = bits(a.cr01 + b.cr01), cr23 = bits(a.cr23 + b.cr23);
            u16x2 cbs = pk(perm(cb23, cb01, 0x05040100u)) + pk(perm(cb23, cb01, 0x07060302u));
            u16x2 crs = pk(perm(cr23, cr01, 0x05040100u)) + pk(perm(cr23, cr01, 0x07060302u));
            *(uint32_t *)(lds + G::cb_off + p * 512 + 4 * g) = bits(cbs);
            *(uint32_t *)(lds + G::cr_off + p * 512 + 4 * g) = bits(crs);
        }
    } else if (MODE == M444) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int item = k * kThreads + tid, g = item & 127, r = item >> 7;
            Row4 a = color_row4(L.in[k * 3], L.in[k * 3 + 1], L.in[k * 3 + 2]);
            *(uint32_t *)(lds + G::y_off + r * kPitch + 4 * g) = a.y4;
            *(uint32_t *)(lds + G::cb_off + r * kPitch + 4 * g) =
                perm(bits(a.cb23), bits(a.cb01), 0x06040200u);
            *(uint32_t *)(lds + G::cr_off + r * kPitch + 4 * g) =
                perm(bits(a.cr23), bits(a.cr01), 0x06040200u);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            int item = k * kThreads + tid, g = item & 127, r = item >> 7;
            *(uint32_t *)(lds + r * kPitch + 4 * g) = L.in[k];
        }
    }
}

// ---- phase B.1: planar LDS -> 64 level-shifted f32 per block lane -------------------
PIXO_DEV void bytes8_to_f32(u32x2 w, float shift, float *v)
{
    v[0] = (float)(w.x & 0xFF) - shift;         v[1] = (float)((w.x >> 8) & 0xFF) - shift;
    v[2] = (float)((w.x >> 16) & 0xFF) - shift; v[3] = (float)(w.x >> 24) - shift;
    v[4] = (float)(w.y & 0xFF) - shift;         v[5] = (float)((w.y >> 8) & 0xFF) - shift;
    v[6] = (float)((w.y >> 16) & 0xFF) - shift; v[7] = (float)(w.y >> 24) - shift;
}

// returns the quantiser class of this lane's block: 0 luminance, 1 chrominance, -1 idle
template <int MODE> PIXO_DEV int phase_fetch(int tid, const uint8_t *lds, Lane<MODE> &L)
{
    typedef Geo<MODE> G;
    const int wave = uniform_i32(tid >> 6), lane = tid & 63;
    if (MODE == M420) {
        if (wave < 2) {
            int m = wave * 16 + (lane >> 2), s = lane & 3;
            const uint8_t *base = lds + G::y_off + ((s >> 1) * 8) * kPitch + m * 16 + (s & 1) * 8;
#pragma unroll
            for (int r = 0; r < 8; r++)
                bytes8_to_f32(*(const u32x2 *)(base + r * kPitch), 128.0f, &L.v[r * 8]);
            return 0;
        }
        if (wave == 2) {
            const uint8_t *base = lds + G::cb_off + (lane >> 5) * 4096 + (lane & 31) * 16;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                u32x4 w = *(const u32x4 *)(base + r * 512);
                // mean of four u8 chroma values, f32: (sum + 4) * 0.25 - 128 == sum*0.25 - 127
                // (jpeg/mod.rs:1650-1653; every value is a multiple of 0.25 below 256: exact)
                float *v = &L.v[r * 8];
                v[0] = (float)(w.x & 0xFFFF) * 0.25f - 127.0f; v[1] = (float)(w.x >> 16) * 0.25f - 127.0f;
                v[2] = (float)(w.y & 0xFFFF) * 0.25f - 127.0f; v[3] = (float)(w.y >> 16) * 0.25f - 127.0f;
                v[4] = (float)(w.z & 0xFFFF) * 0.25f - 127.0f; v[5] = (float)(w.z >> 16) * 0.25f - 127.0f;
                v[6] = (float)(w.w & 0xFFFF) * 0.25f - 127.0f; v[7] = (float)(w.w >> 16) * 0.25f - 127.0f;
            }
            return 1;
        }
        return -1;
    } else if (MODE == M444) {
        if (wave < 3) {
            const uint8_t *base = lds + wave * (8 * kPitch) + lane * 8;
            const float shift = wave == 0 ? 128.0f : 127.0f; // chroma bytes hold C-1
#pragma unroll
            for (int r = 0; r < 8; r++)
                bytes8_to_f32(*(const u32x2 *)(base + r * kPitch), shift, &L.v[r * 8]);
            return wave == 0 ? 0 : 1;
        }
        return -1;
    } else {
        const uint8_t *base = lds + (wave * 8) * kPitch + lane * 8;
#pragma unroll
        for (int r = 0; r < 8; r++)
            bytes8_to_f32(*(const u32x2 *)(base + r * kPitch), 128.0f, &L.v[r * 8]);
        return 0;
    }
}

// ---- phase B.2: f32 AAN DCT, dct.rs:651-700, operation for operation ---------------
#define PIXO_A1 0.70710678118654752440f
#define PIXO_A2 0.5411961f
#define PIXO_A4 1.3065629f
#define PIXO_A5 0.38268343f

PIXO_DEV void aan8(float &d0, float &d1, float &d2, float &d3, float &d4, float &d5, float &d6,
                   float &d7)
{
    float t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6;
    float t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;

    float e0 = t0 + t3, e3 = t0 - t3, e1 = t1 + t2, e2 = t1 - t2;
    float r0 = e0 + e1, r4 = e0 - e1;
    float z1 = (e2 + e3) * PIXO_A1;
    float r2 = e3 + z1, r6 = e3 - z1;

    float o0 = t4 + t5, o1 = t5 + t6, o2 = t6 + t7;
    float z5 = (o0 - o2) * PIXO_A5;
    float z2 = o0 * PIXO_A2 + z5;
    float z4 = o2 * PIXO_A4 + z5;
    float z3 = o1 * PIXO_A1; // A3 == A1
    float z11 = t7 + z3, z13 = t7 - z3;
    float r5 = z13 + z2, r3 = z13 - z2, r1 = z11 + z4, r7 = z11 - z4;

    d0 = r0 * 0.3535534f; d1 = r1 * 0.2548978f; d2 = r2 * 0.2705981f; d3 = r3 * 0.3006724f;
    d4 = r4 * 0.3535534f; d5 = r5 * 0.4499881f; d6 = r6 * 0.6532815f; d7 = r7 * 1.2814578f;
}

PIXO_DEV void dct_2d(float *v)
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        aan8(v[r * 8], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3], v[r * 8 + 4], v[r * 8 + 5],
             v[r * 8 + 6], v[r * 8 + 7]);
        if (r & 1) PIXO_SCHED_FENCE();
    }
#pragma unroll
    for (int c = 0; c < 8; c++) {
        aan8(v[c], v[8 + c], v[16 + c], v[24 + c], v[32 + c], v[40 + c], v[48 + c], v[56 + c]);
        if (c & 1) PIXO_SCHED_FENCE();
    }
}

// ---- phase B.3: quantise one row of 8 coefficients ----------------------------------
// Reference: (x / q).round() as i16 with IEEE f32 divide and round-half-away.
//
// Fast path: r = x * fl(1/q), n = rint(r).  Let t = x/q (real) and f = fl(t) the
// reference quotient.  |r - t| <= |t|(2^-24 + 2^-24 + 2^-48) and |f - t| <= 2^-24|t|,
// so |r - f| < 2^-22 |r| =: delta.  If no half-integer lies within delta' = 2^-21 |r|
// (twice delta; the slack absorbs the rounding of the test itself) of r, then r and f
// sit strictly inside the same interval (k-1/2, k+1/2) and both roundings — rint for
// r, half-away for f — give k.  Otherwise the lane takes the exact divide.  |r - n|
// is computed exactly (Sterbenz).  Tiny |r| (< 1/8) are trivially safe.
PIXO_DEV void quant_row8(const float *x, const float *rcp, const float *q, uint32_t out[4])
{
    float n[8];
    bool risky = false;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        float r = x[c] * rcp[c];
        n[c] = __builtin_rintf(r);
        float lim = __builtin_fmaf(__builtin_fabsf(r), -0x1p-21f, 0.5f);
        risky |= __builtin_fabsf(r - n[c]) >= lim;
    }
    if (risky) { // rare: a quotient within 2^-21 (relative) of a rounding boundary
#pragma unroll
        for (int c = 0; c < 8; c++) {
            n[c] = __builtin_roundf(x[c] / q[c]); // the reference operation itself
            PIXO_SCHED_FENCE();                   // one divide at a time: few temporaries
        }
    }
    out[0] = pack_i16(n[0], n[1]); out[1] = pack_i16(n[2], n[3]);
    out[2] = pack_i16(n[4], n[5]); out[3] = pack_i16(n[6], n[7]);
}

// LDS stage: block b occupies bytes [128b, 128b+128); its 16-byte chunk j (= natural-
// order row j of the block) is stored at slot j ^ (b & 7) so that both the per-block
// ds_write_b128 (8-lane groups, stride 128 B) and the linear ds_read_b128 of phase C
// are bank-conflict free.
PIXO_DEV int stage_addr(int b, int j) { return b * 128 + ((j ^ (b & 7)) << 4); }

template <int MODE>
PIXO_DEV void phase_dct_quant(int tid, int cls, const float *qt, Lane<MODE> &L, uint8_t *lds)
{
    if (cls < 0) return;
    const int wave = uniform_i32(tid >> 6), lane = tid & 63;
    const int b = wave * 64 + lane;
    const float *rcp = qt + uniform_i32(cls) * 64;
    const float *q = rcp + 128;
    dct_2d(L.v);
#pragma unroll
    for (int i = 0; i < 64; i++) PIXO_PIN(L.v[i]);
#pragma unroll
    for (int u = 0; u < 8; u++) {
        u32x4 o;
        uint32_t w[4];
        quant_row8(&L.v[u * 8], rcp + u * 8, q + u * 8, w);
        o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
        *(u32x4 *)(lds + stage_addr(b, u)) = o;
        PIXO_SCHED_FENCE();
    }
}

// ---- phase C: stage -> global, 16 B per lane, coalesced ------------------------------
template <int MODE>
PIXO_DEV void phase_store(const TileCtx &c, uint32_t tile_x, uint32_t tile_y, int tid,
                          const uint8_t *lds)
{
    typedef Geo<MODE> G;
    const uint32_t u0 = tile_x * G::units_x; // first MCU / block column of the tile
    const uint32_t nvalid = c.units_x - u0 < (uint32_t)G::units_x ? c.units_x - u0 : G::units_x;
    if (MODE == M420) {
        const size_t mcu0 = (size_t)tile_y * c.units_x + u0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            int ch = k * kThreads + tid, b = ch >> 3, j = ch & 7;
            u32x4 w = *(const u32x4 *)(lds + stage_addr(b, j));
            if (k < 4) {
                if ((uint32_t)(b >> 2) < nvalid)
                    *(u32x4 *)((uint8_t *)(c.y + mcu0 * 256) + (size_t)ch * 16) = w;
            } else {
                int bi = b & 31;
                int16_t *dst = (k == 4 ? c.cb : c.cr) + (mcu0 + bi) * 64 + j * 8;
                if ((uint32_t)bi < nvalid) *(u32x4 *)dst = w;
            }
        }
    } else if (MODE == M444) {
        const size_t blk0 = (size_t)tile_y * c.units_x + u0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            int ch = k * kThreads + tid, b = ch >> 3, j = ch & 7;
            u32x4 w = *(const u32x4 *)(lds + stage_addr(b, j));
            int bi = b & 63;
            int16_t *plane = (k < 2) ? c.y : (k < 4 ? c.cb : c.cr);
            if ((uint32_t)bi < nvalid) *(u32x4 *)(plane + (blk0 + bi) * 64 + j * 8) = w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int ch = k * kThreads + tid, b = ch >> 3, j = ch & 7;
            u32x4 w = *(const u32x4 *)(lds + stage_addr(b, j));
            uint32_t brow = tile_y * 4 + (b >> 6);
            int bi = b & 63;
            if ((uint32_t)bi < nvalid && brow < c.units_y)
                *(u32x4 *)(c.y + ((size_t)brow * c.units_x + u0 + bi) * 64 + j * 8) = w;
        }
    }
}

} // namespace pixo_tile
