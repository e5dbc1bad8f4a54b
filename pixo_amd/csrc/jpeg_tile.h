// jpeg_tile.h — the per-tile body of the fused JPEG coefficient kernel for gfx950.
//
// One 192-thread workgroup (3 wavefronts) turns one 512-pixel-wide tile of pixels into quantised
// DCT blocks (a tile holds exactly 3 x 64 blocks):
//
//   phase A  (3 wavefronts share the tile's work items 6/5/5)   global RGB8 -> registers: one
//            unconditional 12-byte load per lane and row (4 px; a wavefront instruction covers 768
//            contiguous bytes; rows that are not dword aligned: aligned dwords + v_alignbyte), integer
//            BT.601 colour conversion with packed-u16 VALU ops (2 px per instruction), 2x2 chroma box
//            sums, planar u8/u16 samples into LDS.
//   --- one LDS-only barrier ---
//   phase B  (3 wavefronts, one lane per 8x8 block)  LDS rows -> f32, f32 AAN DCT rows then columns
//            entirely in registers (no transposes), quantise (two bracketing reciprocal products
//            proven to agree with the IEEE divide, exact divide fallback), pack the whole block to
//            32 registers of i16 pairs, stage 32 blocks at a time in the wavefront's own LDS area,
//            read back by 16-byte chunk and store to HBM so that 8 consecutive lanes write one
//            128-byte block, in the reference's YCbCrCoefficients layout.
//
// (Function names say producer_* / consumer_* because the same pieces were also run as
// role-specialised wavefronts of a persistent workgroup; see DESIGN.md for that experiment.)
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
//
// Instruction selection follows tools/ubench/valu_rates.hip measured on MI355X: wave64 VOP2
// v_add/sub/mul_f32, v_and/or/xor/add_u32 issue in ~2 cycles; v_fma_f32, anything with an SGPR
// operand, v_cvt_*, v_cmp, v_perm, v_pk_*, SDWA forms and every VOP3 integer op in ~4.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(PIXO_EMU)
#include <math.h>
#include <string.h>
#define PIXO_DEV static inline
#define PIXO_SCHED_FENCE() ((void)0)
#define PIXO_WAVE_SYNC() ((void)0)
#define PIXO_PIN(x) ((void)0)
#define PIXO_PIN2(x) ((void)0)
#define PIXO_CONST_AS
#else
#define PIXO_DEV __device__ __forceinline__
// Stops the machine scheduler from interleaving independent 1-D transforms (which keeps the
// temporaries of many rows/columns alive at once).
#define PIXO_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// Lanes of a wavefront exchange data through LDS without a workgroup barrier (stage written by block,
// read back by chunk): the hardware runs them in lockstep and keeps a wavefront's LDS operations in
// order, but the COMPILER reasons per thread — without this fence it forwarded a value a lane had
// loaded earlier past stores that only OTHER lanes execute (hipcc 7.2, seen on the 4:4:4 kernel).
// Wavefront-scope release/acquire: no instruction, only the ordering.
#define PIXO_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// Materialises a value here: stops LLVM from sinking the column pass into the quantiser's
// basic blocks (which kept every column's butterflies alive across them).
#define PIXO_PIN(x) asm volatile("" : "+v"(x))
// The same for two floats that must live in ONE aligned register pair (the operand form of v_pk_add_f32 / v_pk_mul_f32).
#define PIXO_PIN2(x) asm volatile("" : "+v"(x))
// The quantiser tables are read through the constant address space so that the compiler may
// use scalar (SMEM) loads.  Inside the persistent loop a plain global pointer is "clobbered"
// by the kernel's own stores as far as alias analysis knows, and becomes VECTOR loads — whose
// vmcnt wait is in-order and therefore also waits for the next tile's prefetch: no overlap.
#define PIXO_CONST_AS __attribute__((address_space(4)))
#endif

#pragma clang fp contract(off)

namespace pixo_tile {

enum Mode { M420 = 0, M444 = 1, MGRAY = 2 };

constexpr int kWaves = 3;         // wavefronts per workgroup = per tile (a tile has 3 x 64 blocks)
constexpr int kThreads = 64 * kWaves;
constexpr int kTileW = 512;     // pixels per tile row
constexpr int kPitch = 528;     // planar row pitch, full-width planes (4:4:4, gray)
constexpr int kPitchHalf = 272; // planar row pitch, 256-px half planes (4:2:0 luminance);
                                // 8*272 % 256 == 128 puts the bottom Y blocks of an MCU on the
                                // other half of the 64 banks: ds_read_b64 conflict-free

// Quantiser table block for one quality, resident in HBM, read with scalar loads (rlo / rhi: the
// bracketing reciprocals of quant_row8, correctly directed roundings made by the host):
//   [0,128)   (rlo, rhi) luminance, coefficient i at 2 i, 2 i + 1 — the pair is ONE operand of the packed multiply-add
//   [128,192) q   luminance     [192,256)  q   chrominance   (f32, exact integers 1..255)
//   [256,384) (rlo, rhi) chrominance
//   [384,512) (rlo, rhi) chrominance/4 — for 4:2:0 chroma, whose DCT runs on
//             the 2x2 SUMS (4x the sample; a power-of-two scale commutes with every f32 rounding)
constexpr int kQtFloats = 512;

// Tiles and planar layouts (bytes inside one planar buffer):
//   4:2:0  512x16 px = 32 MCUs = 192 blocks.  Luminance as two 256-px half planes (16 rows x
//          272 B) at 0 and 4352 — consumer 0 takes MCUs 0-15, consumer 1 MCUs 16-31; 2x2 chroma
//          sums as u16, 8 rows x 512 B: Cb at 8704, Cr at 12800 — consumer 2 takes 32 Cb + 32 Cr.
//   4:4:4  512x8 px = 64 block columns x {Y,Cb,Cr}: plane c (8 rows x 528 B) at 4224 c, consumer c.
//   gray   512x24 px = 3 block rows x 64: block row w (8 rows x 528 B) at 4224 w, consumer w.
// Producer work item k of a lane: 4 px at x = 4 g, g = 64 (k & 1) + lane, image row(s) k >> 1
// (a row PAIR for 4:2:0 so that a lane owns whole chroma quads): consecutive lanes read
// consecutive 12-byte groups, a wavefront instruction covers 768 contiguous bytes of a row.
template <int MODE> struct Geo;
template <> struct Geo<M420> { static constexpr int tile_h = 16, units_x = 32, bpp = 3, items = 16, item_regs = 6, planar = 16896; };
template <> struct Geo<M444> { static constexpr int tile_h = 8, units_x = 64, bpp = 3, items = 16, item_regs = 3, planar = 12672; };
template <> struct Geo<MGRAY> { static constexpr int tile_h = 24, units_x = 64, bpp = 1, items = 48, item_regs = 1, planar = 12672; };
template <int MODE> constexpr int lds_bytes() { return Geo<MODE>::planar; }
// A consumer wavefront's 4 KiB stage reuses the start of the planar area only it reads:
// 4:2:0 luminance half planes (4352 B each) and the chroma sums (8192 B); 4:4:4 / gray planes
// (4224 B each).  Lanes of a wavefront run in lockstep and LDS operations of a wavefront complete
// in order, so all of its planar reads (row pass) precede its first stage write (quantiser).
template <int MODE> PIXO_DEV int stage_offset(int wave) { return MODE == M420 ? wave * 4352 : wave * 4224; }

// Per-image launch context (uniform across the workgroup).
struct TileCtx {
    const uint8_t *px;    // this image's pixels, tightly packed rows
    int16_t *y, *cb, *cr; // this image's coefficient arrays
    const float *qt;      // kQtFloats floats for the requested quality
    uint32_t W, H;        // pixels
    uint32_t units_x;     // MCUs per row (4:2:0) or 8x8 blocks per row (4:4:4, gray)
    uint32_t units_y;     // MCU rows / block rows
    uint32_t fast;        // vector loads (L_ALIGNED or L_FUNNEL); 0 = byte gathers
    const uint8_t *px_end; // one past the last pixel byte of the launch (all images of a batch)
#if defined(PIXO_EMU)
    const uint8_t *px_first; // first pixel byte of the launch (the emulation checks every vector load against [px_first, px_end): emu_check_load)
#endif
};

// How phase A reads the pixels (a launch-time choice, template parameter of the kernel):
//   L_ALIGNED  every row dword aligned (base % 4 == 0, W * bpp % 4 == 0): one 12-byte load per lane and row
//   L_FUNNEL   any alignment, W >= 4: the same 12-byte load at the lane's own byte address (unaligned_load; the name is
//              round 2's, which funnelled aligned dwords through v_alignbyte)
//   L_BYTES    byte gathers (images narrower than one 4-pixel group)
enum Load { L_BYTES = 0, L_ALIGNED = 1, L_FUNNEL = 2 };

// ---------------------------------------------------------------------------------
// small intrinsic wrappers (device instruction / host emulation)
// ---------------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

PIXO_DEV uint32_t bits(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
PIXO_DEV uint32_t fbits(float v) { return __builtin_bit_cast(uint32_t, v); }
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

// v_alignbyte_b32: ({hi, lo} >> 8 * (sh & 3)) & 0xffffffff
PIXO_DEV uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh)
{
#if defined(PIXO_EMU)
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3)));
#else
    return __builtin_amdgcn_alignbyte(hi, lo, sh & 3);
#endif
}

typedef PIXO_CONST_AS const float *qtab_t;
PIXO_DEV qtab_t as_qtab(const float *p) { return (qtab_t)(uintptr_t)p; }

PIXO_DEV int uniform_i32(int v)
{
#if defined(PIXO_EMU)
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}

struct u32x2 { uint32_t x, y; };
// Coefficient stores are non-temporal (global_store_dwordx4 ... nt): the 50 MB a 4096x4096 image
// produces are never read back by this kernel, and written with the default policy ~30 MB of them
// are still dirty in the eight 4 MB L2s when the kernel ends — the end-of-kernel write-back then
// adds ~3.5 us during which nothing else runs (29.2 -> 25.9 us with nt, measured).
#if defined(PIXO_EMU)
#define PIXO_GSTORE(ptr, val) (*(u32x4 *)(ptr) = (val))
#else
typedef uint32_t pixo_v4u __attribute__((ext_vector_type(4)));
#define PIXO_GSTORE(ptr, val) __builtin_nontemporal_store(__builtin_bit_cast(pixo_v4u, (val)), (pixo_v4u *)(ptr))
#endif
struct alignas(16) u32x4 { uint32_t x, y, z, w; };

// ---------------------------------------------------------------------------------
// phase A: colour conversion of 4 horizontally adjacent pixels held as 3 dwords
//   d0 = R0 G0 B0 R1   d1 = G1 B1 R2 G2   d2 = B2 R3 G3 B3   (little-endian bytes)
// ---------------------------------------------------------------------------------
// a*b + c on two u16 lanes, SATURATING at 0xFFFF (v_pk_mad_u16 with the clamp bit).
PIXO_DEV u16x2 mad_sat(u16x2 a, u16x2 b, u16x2 c)
{
#if defined(PIXO_EMU)
    u16x2 o;
    for (int i = 0; i < 2; i++) {
        const uint32_t t = (uint32_t)a[i] * b[i] + c[i];
        o[i] = (unsigned short)(t > 0xFFFFu ? 0xFFFFu : t);
    }
    return o;
#else
    u16x2 o;
    asm("v_pk_mad_u16 %0, %1, %2, %3 clamp" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
#endif
}

// a*k + c on two u16 lanes, wrapping; k is a wave-uniform constant (scalar register).  Written
// as the instruction itself: left to the compiler, u16 arithmetic is reassociated into
// multiply + multiply-add + add chains (one more half-rate instruction per chain).
PIXO_DEV u16x2 mad_k(u16x2 a, unsigned k, u16x2 c)
{
#if defined(PIXO_EMU)
    return a * splat(k) + c;
#else
    u16x2 o;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(o) : "v"(a), "s"(k * 0x10001u), "v"(c));
    return o;
#endif
}

// (high byte of e.lo + high byte of o.lo, high byte of e.hi + high byte of o.hi) as two u16
// lanes: two SDWA adds with byte selects instead of two packed shifts and a packed add.  The
// s_nop's cover the SDWA partial-write forwarding hazard, which the compiler cannot see inside
// inline assembly.
PIXO_DEV u16x2 add_high_bytes(u16x2 e, u16x2 o)
{
#if defined(PIXO_EMU)
    return (e >> 8) + (o >> 8);
#else
    uint32_t d;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n\t"
        "s_nop 0\n\t"
        "v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_3\n\t"
        "s_nop 0"
        : "=&v"(d) : "v"(e), "v"(o));
    return pk(d);
#endif
}

// Pixels are paired even/odd: lane pair E = (pixel 0, pixel 2), O = (pixel 1, pixel 3), so that
// the horizontal neighbours a 4:2:0 box sum adds sit in the same u16 lane of E and O.
struct Row4 {
    uint32_t y4;    // Y0..Y3 as bytes
    u16x2 cbE, cbO; // u16 per pixel whose HIGH byte is Cb (low byte: discarded fraction)
    u16x2 crE, crO;
};

// color.rs:60-77 restated for packed u16 lanes.
//   Y  = (77R + 150G + 29B + 128) >> 8            max 65408: fits u16, clamp is a no-op.
//   Cb = clamp(((-43R - 85G + 128B + 128) >> 8) + 128, 0, 255)
//      = min((X >> 8), 255)  with  X = 128B + I,  I = 32896 - 43R - 85G  in [256, 32896].
//        I is formed with wrapping u16 multiply-adds (multipliers 65536-43, 65536-85: the
//        true value is in range, so the wrapped result is it).  128B + I <= 65536 and reaches
//        65536 only for B = 255, R = G = 0 — exactly the one colour the reference clamps
//        (256 -> 255): the last multiply-add SATURATES, 0xFFFF >> 8 = 255.  No min, no +-1.
//   Cr = same with X = 128R + I,  I = 32896 - 107G - 21B.
// Arithmetic >> on negative i32 in the reference equals the floor that the biased unsigned
// high byte takes.
PIXO_DEV Row4 color_row4(uint32_t d0, uint32_t d1, uint32_t d2)
{
    u16x2 rE = pk(perm(d1, d0, 0x0C060C00u)); // R0 R2
    u16x2 gE = pk(perm(d1, d0, 0x0C070C01u)); // G0 G2
    u16x2 bE = pk(perm(d2, d0, 0x0C040C02u)); // B0 B2
    u16x2 rO = pk(perm(d2, d0, 0x0C050C03u)); // R1 R3
    u16x2 gO = pk(perm(d2, d1, 0x0C060C00u)); // G1 G3
    u16x2 bO = pk(perm(d2, d1, 0x0C070C01u)); // B1 B3

    // (a VOP3P instruction may read one scalar operand: the two constant addends live in VGPRs)
    u16x2 k128 = splat(128), kI = splat(32896);
    PIXO_PIN(k128); PIXO_PIN(kI);
    const unsigned kM43 = 65536 - 43, kM85 = 65536 - 85, kM107 = 65536 - 107, kM21 = 65536 - 21;
    u16x2 yE = mad_k(rE, 77, mad_k(gE, 150, mad_k(bE, 29, k128)));
    u16x2 yO = mad_k(rO, 77, mad_k(gO, 150, mad_k(bO, 29, k128)));
    Row4 o;
    o.y4 = perm(bits(yO), bits(yE), 0x07030501u); // high byte of each u16 lane, pixel order
    o.cbE = mad_sat(bE, k128, mad_k(rE, kM43, mad_k(gE, kM85, kI)));
    o.cbO = mad_sat(bO, k128, mad_k(rO, kM43, mad_k(gO, kM85, kI)));
    o.crE = mad_sat(rE, k128, mad_k(gE, kM107, mad_k(bE, kM21, kI)));
    o.crO = mad_sat(rO, k128, mad_k(gO, kM107, mad_k(bO, kM21, kI)));
    return o;
}

// ---- the same conversion with dot-product instructions (one instruction = one pixel's three multiply-adds) ----------
// For 4:2:0, where the chroma values are only ever needed as bytes to add up.  A pixel's three bytes lie in ONE dword
// (pixels 0 and 3 of a group do already: d0 = R0 G0 B0 R1, d2 = B2 R3 G3 B3; pixels 1 and 2 after one v_alignbyte
// each); the coefficient dword says which byte is which, so pixel 3 needs no shift.
//   Y = 77 R + 150 G + 29 B + 128 <= 65408: v_dot4_u32_u8 on the pixel as it is, the value is byte 1.
// Chroma has negative coefficients.  Round 5: ONE xor per pixel turns its bytes into SIGNED ones (s = x - 128) and
// v_dot4_i32_i8 takes them with signed coefficients — the offsets cancel, (-43 - 85 + 128) * 128 = 0:
//   Cb:  X = 128 B - 43 R - 85 G + 32896 = 128 sB - 43 sR - 85 sG + 32896     in [256, 65536]
//   Cr:  X = 128 R - 107 G - 21 B + 32896 = 128 sR - 107 sG - 21 sB + 32896
// +128 is not an i8 — but -128 is: the lane computes the COMPLEMENT  255 - Cb = 255 - min(X >> 8, 255) = max(V, 0) >> 8
// with V = 65535 - X = 43 sR + 85 sG - 128 sB + 32639 in [-1, 65279] (the identity 255 - floor(X / 256) =
// floor((65535 - X) / 256) holds for 0 <= X <= 65535).  V = -1 is X = 65536 — pure blue / pure red, the one colour the
// reference clamps (256 -> 255, color.rs:69-76): the accumulator starts at 32639 - 2^31 and the instruction SATURATES at
// -2^31, so that the result is 0x80000000 + max(V, 0) and byte 1 is the complemented value.  The 2x2 sums are then
// S' = 1020 - S, every value of the chroma transform the exact negative of the reference's (all inputs are integers,
// f32 rounding is symmetric), and the quantiser multiplies by NEGATED reciprocals (jpeg_host.cpp fill_device_qt): the same
// coefficients.  (Rounds 3-4 complemented the bytes with negative coefficients: two xors per pixel.)
// 12 dot products + 4 xors + 2 shifts per 4 pixels against 18 packed multiply-adds + 6 byte shuffles; the price is that
// every result is a dword of its own (three more instructions to gather the four Y bytes).
PIXO_DEV uint32_t udot4(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(PIXO_EMU)
    uint32_t t = c;
    for (int i = 0; i < 4; i++) t += ((a >> (8 * i)) & 0xFF) * ((b >> (8 * i)) & 0xFF);
    return t;
#else
    return __builtin_amdgcn_udot4(a, b, c, false);
#endif
}
// signed bytes, signed 32-bit accumulator, the result saturated to [-2^31, 2^31 - 1] (v_dot4_i32_i8 ... clamp)
PIXO_DEV uint32_t sdot4_sat(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(PIXO_EMU)
    int64_t t = (int32_t)c;
    for (int i = 0; i < 4; i++) t += (int64_t)(int8_t)(a >> (8 * i)) * (int8_t)(b >> (8 * i));
    t = t < -2147483648ll ? -2147483648ll : (t > 2147483647ll ? 2147483647ll : t);
    return (uint32_t)(int32_t)t;
#else
    return (uint32_t)__builtin_amdgcn_sdot4((int)a, (int)b, (int)c, true);
#endif
}
// What the chroma planes of the dot-product path hold, and what phase B therefore uses for a 4:2:0 chroma block:
constexpr float kChromaSumShift = 8.0f * 508.0f; // row DC shift: 512 - S = -(S' - 508), S' = 1020 - S
constexpr float kChromaSumScale = -0.25f;        // back to the reference's magnitude AND sign
struct Row4D {
    uint32_t y4;           // Y0..Y3 as bytes
    uint32_t cb[4], cr[4]; // per pixel: byte 1 = 255 - the value (byte 0: discarded fraction, bytes 2-3: 0x00 0x80)
};
PIXO_DEV Row4D color_row4_dot(uint32_t d0, uint32_t d1, uint32_t d2)
{
    const uint32_t p[4] = {d0, alignbyte(d1, d0, 3), alignbyte(d2, d1, 2), d2}; // pixel i at bytes 0..2 (i = 3: bytes 1..3)
    const uint32_t kC = 0x80000000u + 32639u;
    uint32_t y[4];
    Row4D o;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int sh = i == 3 ? 8 : 0;
        y[i] = udot4(p[i], 0x001D964Du << sh, 128u);             // 77, 150, 29
        const uint32_t s = p[i] ^ (0x00808080u << sh);           // signed bytes
        o.cb[i] = sdot4_sat(s, 0x0080552Bu << sh, kC);           // 43 sR, 85 sG, -128 sB
        o.cr[i] = sdot4_sat(s, 0x00156B80u << sh, kC);           // -128 sR, 107 sG, 21 sB
    }
    o.y4 = perm(y[1], y[0], 0x0C0C0501u) | perm(y[3], y[2], 0x05010C0Cu); // byte 1 of each, pixel order
    return o;
}
// (byte 1 of a + byte 1 of b, byte 1 of c + byte 1 of d) as two u16 lanes (see add_high_bytes)
PIXO_DEV uint32_t add_bytes1(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
#if defined(PIXO_EMU)
    return (((a >> 8) & 0xFF) + ((b >> 8) & 0xFF)) | ((((c >> 8) & 0xFF) + ((d >> 8) & 0xFF)) << 16);
#else
    uint32_t r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n\t"
        "s_nop 0\n\t"
        "v_add_u32_sdwa %0, %3, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_1\n\t"
        "s_nop 0"
        : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
    return r;
#endif
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

// ---------------------------------------------------------------------------------
// producer: the wavefronts move a whole tile from HBM to a planar LDS buffer
// ---------------------------------------------------------------------------------
// The vector paths read with unconditional loads and no branch anywhere near a load: the address
// is clamped to the last 4-pixel group of the row (x = W - 4) and to the last row, which IS the
// reference's edge replication for rows (jpeg/mod.rs:1579,1627); a group that reaches beyond the
// image is rebuilt from the clamped group's pixels by fix_right_edge_*, a register-only step that
// only right-edge tiles execute.  (Branches around the loads made hipcc wait for every load at its
// join point: one outstanding load per wavefront, 4.6 TB/s at best.)
//
// Pixels are read once: the loads are non-temporal (they do not displace the other workgroups'
// lines in L2 / Infinity Cache on their way through).
#if defined(PIXO_EMU)
#define PIXO_GLOAD(ptr) (*(ptr))
#else
#define PIXO_GLOAD(ptr) __builtin_nontemporal_load(ptr)
#endif

// L_UNALIGNED: `N` (1 or 3) dwords starting at byte address `row + off`, whatever its alignment, with ONE load at that
// address: global memory is in unaligned access mode under ROCm (the compiler itself emits global_load_dwordx3 for a vector
// of alignment 1), and exactly the lane's own bytes are read — nothing before the row, nothing behind the buffer.
// Round 2 read the N + 1 ALIGNED dwords around them and funnelled those through v_alignbyte (three half-rate operations
// per row + address arithmetic, a special case for the tiles that hold the buffer's last dword): 20.8-21.2 us for a
// 4094-pixel-wide image against 19.3-19.6 now (aligned 4096: 18.3; profiles/r03_unaligned_direct_loads.txt).
#if defined(PIXO_EMU)
// (host emulation only) every vector load of the pixel path must lie inside the launch's pixel bytes: the tests then catch a
// load that the device would get away with.  Set by the harness around a launch (tests/emu/emu_tile.cpp).
static const uint8_t *g_emu_px_first = nullptr, *g_emu_px_end = nullptr;
static long g_emu_oob_loads = 0;
static inline void emu_check_load(const uint8_t *p, size_t n)
{
    if (g_emu_px_first && (p < g_emu_px_first || p + n > g_emu_px_end)) g_emu_oob_loads++;
}
#endif
template <int N> PIXO_DEV void unaligned_load(const uint8_t *row, uint32_t off, uint32_t *r)
{
#if defined(PIXO_EMU)
    emu_check_load(row + off, 4 * N);
    memcpy(r, row + off, 4 * N);
#else
    if (N == 3) {
        typedef uint32_t v3a1 __attribute__((ext_vector_type(3), aligned(1)));
        const v3a1 q = PIXO_GLOAD((const v3a1 *)(row + off));
        r[0] = q.x; r[1] = q.y; r[2] = q.z;
    } else {
        typedef uint32_t v1a1 __attribute__((aligned(1)));
        r[0] = PIXO_GLOAD((const v1a1 *)(row + off));
    }
#endif
}

// What a lane needs to address its pixels, computed ONCE per wavefront (vector paths): the tile's first row as a
// wave-uniform 64-bit pointer and the lane's byte offset inside a row for each half of the tile — clamped to the row's last
// 4-pixel group, which is the reference's replicate rule.  Every load is then  tile_row0 + (row inside the tile) * stride
// [scalar, 32-bit arithmetic] + xoff[half] [vector]: five or six scalar instructions per row.  The scalar unit is shared
// by the CU's 24 wavefronts, which all start within a microsecond and all do their address arithmetic before their first
// load: with a 64-bit multiplication per row and divisions for the tile's coordinates (~300 scalar instructions per
// wavefront) the median wavefront issued its loads 4.5 us after the dispatch began, the memory system idle meanwhile
// (profiles/r03_timeline_c2_before.txt).
struct LaneAddr {
    const uint8_t *tile_row0; // uniform: first byte of the tile's first pixel row (x = 0)
    uint32_t stride;          // uniform: bytes per pixel row
    uint32_t row_first;       // uniform: the tile's first pixel row
    uint32_t xoff0, xoff1;    // per lane: byte offset of the lane's 4-pixel group in the left / right half of the tile
};
template <int MODE> PIXO_DEV LaneAddr lane_addr(const TileCtx &c, uint32_t tile_x, uint32_t tile_y, int lane)
{
    typedef Geo<MODE> G;
    LaneAddr a;
    a.stride = c.W * (uint32_t)G::bpp;
    a.row_first = tile_y * (uint32_t)G::tile_h;
    a.tile_row0 = c.px + (size_t)a.row_first * a.stride;
    const uint32_t x0 = tile_x * kTileW + 4u * (uint32_t)lane, x1 = x0 + 256u, last = c.W >= 4 ? c.W - 4 : 0u;
    a.xoff0 = (x0 < last ? x0 : last) * (uint32_t)G::bpp;
    a.xoff1 = (x1 < last ? x1 : last) * (uint32_t)G::bpp;
    PIXO_PIN(a.xoff0); PIXO_PIN(a.xoff1); // (two registers for the whole phase; left alone the compiler re-derives them per item from three)
    return a;
}

// three dwords at a dword-aligned address: ONE 12-byte load (three separate loads left the merging to the compiler, which
// at times made a 4-byte and an 8-byte load with a 64-bit vector address of their own)
PIXO_DEV void load_dwordx3(const uint8_t *p, uint32_t *r)
{
#if defined(PIXO_EMU)
    emu_check_load(p, 12);
    memcpy(r, p, 12);
#else
    typedef uint32_t v3a4 __attribute__((ext_vector_type(3), aligned(4)));
    const v3a4 q = PIXO_GLOAD((const v3a4 *)p);
    r[0] = q.x; r[1] = q.y; r[2] = q.z;
#endif
}

template <int MODE, int LOAD>
PIXO_DEV void producer_load_item(const TileCtx &c, const LaneAddr &la, uint32_t tile_x, uint32_t tile_y, int k, int lane,
                                 uint32_t *r)
{
    if (LOAD != L_BYTES) {
        // rows inside the tile, clamped to the image's last row (the reference's replicate rule for rows,
        // jpeg/mod.rs:1579,1627); everything wave-uniform and 32-bit: a tile has at most 24 rows of at most 196,605 bytes
        const uint32_t y0 = la.row_first + (MODE == M420 ? 2u : 1u) * (uint32_t)(k >> 1);
        const uint32_t ya = y0 < c.H ? y0 : c.H - 1;
        const uint8_t *row = la.tile_row0 + (ya - la.row_first) * la.stride;
        const uint32_t xoff = (k & 1) ? la.xoff1 : la.xoff0; // (a select, not an indexed array: that would live in scratch)
        if (MODE == MGRAY) {
#if defined(PIXO_EMU)
            if (LOAD == L_ALIGNED) emu_check_load(row + xoff, 4);
#endif
            if (LOAD == L_ALIGNED) r[0] = PIXO_GLOAD((const uint32_t *)(row + xoff));
            else unaligned_load<1>(row, xoff, r);
        } else {
            if (LOAD == L_ALIGNED) load_dwordx3(row + xoff, r);
            else unaligned_load<3>(row, xoff, r);
            if (MODE == M420) {
                const uint32_t yb = y0 + 1 < c.H ? y0 + 1 : c.H - 1;
                const uint8_t *row2 = la.tile_row0 + (yb - la.row_first) * la.stride;
                if (LOAD == L_ALIGNED) load_dwordx3(row2 + xoff, r + 3);
                else unaligned_load<3>(row2, xoff, r + 3);
            }
        }
    } else {
        const uint32_t x0 = tile_x * kTileW + 4 * ((k & 1) * 64 + lane);
        const uint32_t y0 = tile_y * Geo<MODE>::tile_h + (MODE == M420 ? 2 : 1) * (k >> 1);
        if (MODE == MGRAY) {
            r[0] = gather_row4_gray(c.px, c.W, c.H, x0, y0);
        } else {
            u32x3 t = gather_row4_rgb(c.px, c.W, c.H, x0, y0);
            r[0] = t.a; r[1] = t.b; r[2] = t.c;
            if (MODE == M420) {
                t = gather_row4_rgb(c.px, c.W, c.H, x0, y0 + 1);
                r[3] = t.a; r[4] = t.b; r[5] = t.c;
            }
        }
    }
}

// Right-edge fix-up for the vector loads.  The group was read at x = W - 4 (pixels p0..p3 = W-4..W-1)
// instead of x0 = W - 4 + d, d >= 1; what the reference's clamp (x = min(x, W-1)) yields for it is
// (p[min(d,3)], p[min(d+1,3)], p[min(d+2,3)], p3).  d in {1, 2} occurs only when W % 4 != 0 (L_FUNNEL),
// in one lane per row; d >= 3 is a group wholly right of the image: four copies of p3, bytes 9..11.
template <int LOAD> PIXO_DEV void fix_right_edge_rgb(uint32_t d, uint32_t *v)
{
    const uint32_t c0 = perm(v[2], v[2], 0x01030201u); // R3 G3 B3 R3
    const uint32_t c1 = perm(v[2], v[2], 0x02010302u); // G3 B3 R3 G3
    const uint32_t c2 = perm(v[2], v[2], 0x03020103u); // B3 R3 G3 B3
    if (LOAD == L_ALIGNED) { // W % 4 == 0: a group is inside (d = 0) or wholly outside (d = 3)
        v[0] = d == 0 ? v[0] : c0; v[1] = d == 0 ? v[1] : c1; v[2] = d == 0 ? v[2] : c2;
        return;
    }
    const uint32_t a0 = alignbyte(v[1], v[0], 3), a1 = alignbyte(v[2], v[1], 3); // d = 1: p1 p2 p3 p3
    const uint32_t b0 = alignbyte(v[2], v[1], 2);                                 // d = 2: p2 p3 p3 p3
    v[0] = d == 0 ? v[0] : (d == 1 ? a0 : (d == 2 ? b0 : c0));
    v[1] = d == 0 ? v[1] : (d == 1 ? a1 : c1);
    v[2] = d == 0 ? v[2] : c2;
}

template <int MODE, int LOAD>
PIXO_DEV void producer_fix_item(const TileCtx &c, uint32_t tile_x, int k, int lane, uint32_t *r)
{
    if (LOAD == L_BYTES || (tile_x + 1) * kTileW <= c.W) return; // wave-uniform: only right-edge tiles
    const uint32_t x0 = tile_x * kTileW + 4 * ((k & 1) * 64 + lane);
    uint32_t d = x0 > c.W - 4 ? x0 - (c.W - 4) : 0;
    d = d < 3 ? d : 3;
    if (MODE == MGRAY) {
        const uint32_t s1 = perm(r[0], r[0], 0x03030201u), s2 = perm(r[0], r[0], 0x03030302u), s3 = perm(r[0], r[0], 0x03030303u);
        r[0] = d == 0 ? r[0] : (d == 1 ? s1 : (d == 2 ? s2 : s3));
    } else {
        fix_right_edge_rgb<LOAD>(d, r);
        if (MODE == M420) fix_right_edge_rgb<LOAD>(d, r + 3);
    }
}

// DOT4: the 4:2:0 conversion with v_dot4_u32_u8 (color_row4_dot) — the kernels that read pixels with vector loads; the
// byte-gather kernel (images narrower than four pixels) keeps the packed multiply-adds.
template <int MODE, bool DOT4 = false> PIXO_DEV void producer_color_item(int k, int lane, const uint32_t *r, uint8_t *planar)
{
    const int h = k & 1, g = h * 64 + lane, row = k >> 1;
    if (MODE == M420) {
        uint8_t *yp = planar + h * 4352 + (2 * row) * kPitchHalf + 4 * lane;
        if (DOT4) {
            const Row4D a = color_row4_dot(r[0], r[1], r[2]);
            PIXO_SCHED_FENCE(); // one row at a time: few temporaries
            const Row4D b = color_row4_dot(r[3], r[4], r[5]);
            *(uint32_t *)yp = a.y4;
            *(uint32_t *)(yp + kPitchHalf) = b.y4;
            // 2x2 box sums (jpeg/mod.rs:1641-1646), u16 exact (<= 1020); the two rows' halves never carry into each other
            *(uint32_t *)(planar + 8704 + row * 512 + 4 * g) = add_bytes1(a.cb[0], a.cb[1], a.cb[2], a.cb[3]) + add_bytes1(b.cb[0], b.cb[1], b.cb[2], b.cb[3]);
            *(uint32_t *)(planar + 12800 + row * 512 + 4 * g) = add_bytes1(a.cr[0], a.cr[1], a.cr[2], a.cr[3]) + add_bytes1(b.cr[0], b.cr[1], b.cr[2], b.cr[3]);
            return;
        }
        Row4 a = color_row4(r[0], r[1], r[2]);
        PIXO_SCHED_FENCE(); // one row at a time: few temporaries
        Row4 b = color_row4(r[3], r[4], r[5]);
        *(uint32_t *)yp = a.y4;
        *(uint32_t *)(yp + kPitchHalf) = b.y4;
        // 2x2 box sums (jpeg/mod.rs:1641-1646) of the high bytes, u16 exact (<= 1020):
        // even + odd lanes = horizontal neighbours, then the two rows
        // (stored complemented, S' = 1020 - S, like the dot-product path: phase B knows one convention)
        u16x2 cbs = splat(1020) - (add_high_bytes(a.cbE, a.cbO) + add_high_bytes(b.cbE, b.cbO));
        u16x2 crs = splat(1020) - (add_high_bytes(a.crE, a.crO) + add_high_bytes(b.crE, b.crO));
        *(uint32_t *)(planar + 8704 + row * 512 + 4 * g) = bits(cbs);
        *(uint32_t *)(planar + 12800 + row * 512 + 4 * g) = bits(crs);
    } else if (MODE == M444) {
        Row4 a = color_row4(r[0], r[1], r[2]);
        uint8_t *p = planar + row * kPitch + 4 * g;
        *(uint32_t *)p = a.y4;
        *(uint32_t *)(p + 4224) = perm(bits(a.cbO), bits(a.cbE), 0x07030501u);
        *(uint32_t *)(p + 8448) = perm(bits(a.crO), bits(a.crE), 0x07030501u);
    } else {
        *(uint32_t *)(planar + (row >> 3) * 4224 + (row & 7) * kPitch + 4 * g) = r[0];
    }
}

// ---------------------------------------------------------------------------------
// consumers: one lane = one 8x8 block
// ---------------------------------------------------------------------------------
// Samples (bytes, or 2x2 sums in u16 lanes) become floats with one conversion each
// (v_cvt_f32_ubyteN / v_cvt_f32_u32 with a sub-dword select: half rate, like every other way of
// isolating a byte).  The JPEG level shift needs no per-sample operation: the row transform's
// first butterflies are sums and differences of exact small integers, the shift cancels in
// every difference and reaches only the DC term of the row, where one subtraction removes it
// (same arithmetic as the reference's `x as f32 - 128.0`, all values exact integers in f32).
// Two rows at a time (round 5): sample c of rows r and r + 1 becomes the two halves of ONE aligned register pair, the
// operand form of the packed f32 instructions the row pass now runs on (aan8_pair).
typedef float pixo_cf2 __attribute__((ext_vector_type(2)));
PIXO_DEV void rows2_from_bytes(u32x2 a, u32x2 b, pixo_cf2 *d)
{
    d[0] = (pixo_cf2){(float)(a.x & 0xFF), (float)(b.x & 0xFF)};
    d[1] = (pixo_cf2){(float)((a.x >> 8) & 0xFF), (float)((b.x >> 8) & 0xFF)};
    d[2] = (pixo_cf2){(float)((a.x >> 16) & 0xFF), (float)((b.x >> 16) & 0xFF)};
    d[3] = (pixo_cf2){(float)(a.x >> 24), (float)(b.x >> 24)};
    d[4] = (pixo_cf2){(float)(a.y & 0xFF), (float)(b.y & 0xFF)};
    d[5] = (pixo_cf2){(float)((a.y >> 8) & 0xFF), (float)((b.y >> 8) & 0xFF)};
    d[6] = (pixo_cf2){(float)((a.y >> 16) & 0xFF), (float)((b.y >> 16) & 0xFF)};
    d[7] = (pixo_cf2){(float)(a.y >> 24), (float)(b.y >> 24)};
    // (keeps LLVM from folding the first butterflies into integer SDWA adds + conversions:
    // twice as many half-rate instructions)
#pragma unroll
    for (int i = 0; i < 8; i++) PIXO_PIN2(d[i]);
}
PIXO_DEV void rows2_from_u16(u32x4 a, u32x4 b, pixo_cf2 *d)
{
    d[0] = (pixo_cf2){(float)(a.x & 0xFFFF), (float)(b.x & 0xFFFF)}; d[1] = (pixo_cf2){(float)(a.x >> 16), (float)(b.x >> 16)};
    d[2] = (pixo_cf2){(float)(a.y & 0xFFFF), (float)(b.y & 0xFFFF)}; d[3] = (pixo_cf2){(float)(a.y >> 16), (float)(b.y >> 16)};
    d[4] = (pixo_cf2){(float)(a.z & 0xFFFF), (float)(b.z & 0xFFFF)}; d[5] = (pixo_cf2){(float)(a.z >> 16), (float)(b.z >> 16)};
    d[6] = (pixo_cf2){(float)(a.w & 0xFFFF), (float)(b.w & 0xFFFF)}; d[7] = (pixo_cf2){(float)(a.w >> 16), (float)(b.w >> 16)};
#pragma unroll
    for (int i = 0; i < 8; i++) PIXO_PIN2(d[i]);
}

// f32 AAN DCT, dct.rs:651-700, operation for operation.
#define PIXO_A1 0.70710678118654752440f
#define PIXO_A2 0.5411961f
#define PIXO_A4 1.3065629f
#define PIXO_A5 0.38268343f

// One 1-D transform of dct.rs:651-700 on TWO independent vectors at once — the halves of eight register pairs: rows r and
// r + 1 in the row pass, columns 2 k and 2 k + 1 in the column pass.  Every operation is the reference's, in the reference's
// order, applied to both halves by one packed instruction (v_pk_add_f32 / v_pk_mul_f32: IEEE per half, no contraction), so
// each half sees exactly the roundings of the scalar sequence.  The closing scale multiplications are left to the caller:
// the column pass does them packed, the row pass as plain multiplications that write straight into the halves of the
// COLUMN pass's pairs (neighbouring columns of one row) — the transposition between the two pairings costs nothing.
//
// SHIFT (row pass): the samples come unshifted (true inputs b_i - L).  The butterflies' sums carry +2L, +4L, +8L and the
// differences nothing, so only r0 needs the shift (8L = dc_shift).  Every value up to the first multiplication is an exact
// integer below 2^14, so the reference's own sequence of f32 additions yields the same numbers.
template <bool SHIFT, class T = pixo_cf2>
PIXO_DEV void aan8_pair(T dc_shift, const T *d, T *r)
{
    const T t0 = d[0] + d[7], t7 = d[0] - d[7], t1 = d[1] + d[6], t6 = d[1] - d[6];
    const T t2 = d[2] + d[5], t5 = d[2] - d[5], t3 = d[3] + d[4], t4 = d[3] - d[4];
    const T e0 = t0 + t3, e3 = t0 - t3, e1 = t1 + t2, e2 = t1 - t2;
    r[0] = SHIFT ? (e0 + e1) - dc_shift : e0 + e1;
    r[4] = e0 - e1;
    const T z1 = (e2 + e3) * PIXO_A1;
    r[2] = e3 + z1; r[6] = e3 - z1;
    const T o0 = t4 + t5, o1 = t5 + t6, o2 = t6 + t7;
    const T z5 = (o0 - o2) * PIXO_A5;
    const T z2 = o0 * PIXO_A2 + z5;
    const T z4 = o2 * PIXO_A4 + z5;
    const T z3 = o1 * PIXO_A1; // A3 == A1
    const T z11 = t7 + z3, z13 = t7 - z3;
    r[5] = z13 + z2; r[3] = z13 - z2; r[1] = z11 + z4; r[7] = z11 - z4;
}
// the scale factors that close each pass (dct.rs:689-699)
#define PIXO_S0 0.3535534f
#define PIXO_S1 0.2548978f
#define PIXO_S2 0.2705981f
#define PIXO_S3 0.3006724f
#define PIXO_S4 0.3535534f
#define PIXO_S5 0.4499881f
#define PIXO_S6 0.6532815f
#define PIXO_S7 1.2814578f

// Quantise one row of 8 coefficients.
// Reference: (x / q).round() as i16 with IEEE f32 divide and round-half-away.
//
// Fast path: two reciprocal products that BRACKET the reference quotient.  Let t = x/q (real) and
// f = fl(t) the reference quotient, |f - t| <= 2^-24 |t|.  The table holds, per divisor,
//     rlo = the largest  f32 <= (1/q)(1 - 2^-24 - 2^-30)
//     rhi = the smallest f32 >= (1/q)(1 + 2^-24 + 2^-30)          (jpeg_host.cpp fill_device_qt)
// so that f lies STRICTLY between the real products x*rlo and x*rhi (whatever the sign of x).
//     s_lo = fma(x, rlo, 1.5*2^23)    one rounding of the exact sum: the integer nearest to x*rlo
//     s_hi = fma(x, rhi, 1.5*2^23)    (ties to even) sits, two's complement, in the low mantissa bits
// If both are the same integer n, both products lie in [n - 1/2, n + 1/2], f strictly inside, and
// the reference's round-half-away gives n as well.  If they differ (the quotient is within ~2^-22
// relative of a rounding boundary: a few per cent of wave-rows on noise, far fewer on photographs)
// the row repeats the test per element and takes the reference's own divide only for the elements
// some lane flagged.  Two multiply-adds, one xor, one or per element; the claim is also checked by
// ENUMERATION over every f32 |x| <= 4096 and every q in 1..255 (tests/emu/sweep_quant.py ->
// profiles/quant_fastpath_sweep.txt).  `scale` (1 or 1/4) maps x back to the reference's magnitude
// for the exact path only; the fast path's tables already contain it (exact power of two).
constexpr float kRoundMagic = 12582912.0f; // 1.5 * 2^23

#if defined(PIXO_EMU)
#define PIXO_ANY_LANE(pred) (pred)
#else
#define PIXO_ANY_LANE(pred) (__builtin_amdgcn_ballot_w64(pred) != 0)
#endif

// the two roundings of one element; returns the bits that differ (0 = safe), *s = the low one
PIXO_DEV uint32_t quant_bracket(float x, float rlo, float rhi, float *s)
{
    const float lo = __builtin_fmaf(x, rlo, kRoundMagic);
    const float hi = __builtin_fmaf(x, rhi, kRoundMagic);
    *s = lo;
    return fbits(lo) ^ fbits(hi);
}

// The same with the pair (rlo, rhi) as it lies in the table (round 3): on the device ONE packed multiply-add
// (v_pk_fma_f32: x broadcast to both halves, the pair straight from two scalar registers, the rounding constant from a
// register pair) where two v_fma_f32 with a scalar operand each took an issue slot pair of their own, and the
// difference folded into the row's flag by one three-input bit operation instead of a compare per element —
// 3 issue slots per coefficient instead of 6, of a kernel whose time IS its vector instruction issue.
struct QPair { float lo, hi; };
#if defined(PIXO_EMU)
PIXO_DEV uint32_t quant_bracket_pair(float x, QPair r, float *s) { return quant_bracket(x, r.lo, r.hi, s); }
#else
typedef float pixo_f2 __attribute__((ext_vector_type(2)));
PIXO_DEV uint32_t quant_bracket_pair(float x, QPair r, float *s)
{
    const pixo_f2 p = __builtin_elementwise_fma((pixo_f2){x, x}, (pixo_f2){r.lo, r.hi}, (pixo_f2){kRoundMagic, kRoundMagic});
    *s = p.x;
    return fbits(p.x) ^ fbits(p.y);
}
#endif

// {hi[15:0], lo[15:0]}
PIXO_DEV uint32_t pack_lo16(uint32_t hi, uint32_t lo)
{
    // (v_pack_b32_f16 would be one instruction, but it is a FLOAT operation: it does not pass every bit pattern through —
    // the q = 100 gradient golden fails with it — and it is no faster; profiles/r03_pack_f16_ab.txt)
    return perm(hi, lo, 0x05040100u);
}
// four coefficients -> two registers of packed i16 pairs
// PACKED: the two roundings of a coefficient by ONE v_pk_fma_f32 (round 3) — or, not PACKED, by two v_fma_f32 with a scalar
// operand each.  The packed form issues half as many instructions and is what the issue-bound launches want (4:4:4); on the
// one-generation 4:2:0 launch the plain form is 0.5 us FASTER (19.06 -> 18.53 us, profiles/r05_ab_scalar_vs_packed.txt), like
// the scalar DCT passes: few wavefronts per SIMD at the launch's end, and packed f32 instructions do not pipeline behind each other.
template <bool PACKED>
PIXO_DEV void quant_row4(const float *x, const QPair *r, qtab_t q, float scale, uint32_t out[2])
{
    float s[4];
    uint32_t differ = 0;
#if !defined(PIXO_EMU)
    if (!PACKED) {
#endif
#pragma unroll
        for (int c = 0; c < 4; c++) differ |= quant_bracket(x[c], r[c].lo, r[c].hi, &s[c]);
#if !defined(PIXO_EMU)
        PIXO_PIN(differ);
    }
    pixo_f2 xx[2];
    if (PACKED) {
    // Two neighbouring coefficients share one aligned register pair and each multiply-add broadcasts its own half
    // (op_sel).  The pair is pinned as a pair: left to itself the compiler makes every x the LOW half of a pair of its
    // own — 64 live floats then want 128 registers and the kernel spills 136.
#pragma unroll
    for (int c = 0; c < 4; c += 2) {
        xx[c / 2] = (pixo_f2){x[c], x[c + 1]};
        asm volatile("" : "+v"(xx[c / 2]));
        const pixo_f2 m = {kRoundMagic, kRoundMagic};
        const pixo_f2 p0 = __builtin_elementwise_fma(__builtin_shufflevector(xx[c / 2], xx[c / 2], 0, 0), (pixo_f2){r[c].lo, r[c].hi}, m);
        const pixo_f2 p1 = __builtin_elementwise_fma(__builtin_shufflevector(xx[c / 2], xx[c / 2], 1, 1), (pixo_f2){r[c + 1].lo, r[c + 1].hi}, m);
        s[c] = p0.x; s[c + 1] = p1.x;
        differ |= (fbits(p0.x) ^ fbits(p0.y)) | (fbits(p1.x) ^ fbits(p1.y));
    }
    PIXO_PIN(differ); // (as bit operations — one per coefficient; otherwise it becomes a compare per coefficient again)
    }
#endif
    // low 16 bits of each s = the i16 result (never saturates: |x/q| <= 2^11)
    out[0] = pack_lo16(fbits(s[1]), fbits(s[0]));
    out[1] = pack_lo16(fbits(s[3]), fbits(s[2]));
    const bool flagged = PIXO_ANY_LANE(differ != 0);
    if (flagged) { // rare: some quotient next to a rounding boundary
#pragma unroll
        for (int c = 0; c < 4; c++) {
#if defined(PIXO_EMU)
            const float x0 = x[c];
#else
            const float x0 = PACKED ? ((c & 1) ? xx[c / 2].y : xx[c / 2].x) : x[c];
#endif
            float xc = x0, t;
            PIXO_PIN(xc); // recompute the test here: reusing the fast path's values keeps them all alive
            if (PIXO_ANY_LANE((PACKED ? quant_bracket_pair(xc, r[c], &t) : quant_bracket(xc, r[c].lo, r[c].hi, &t)) != 0)) {
                const float n = __builtin_roundf((x0 * scale) / q[c]); // the reference operation itself
                s[c] = n + kRoundMagic;                                  // exact: |n| < 2^15
            }
            PIXO_SCHED_FENCE(); // one element at a time: few temporaries
        }
        out[0] = pack_lo16(fbits(s[1]), fbits(s[0]));
        out[1] = pack_lo16(fbits(s[3]), fbits(s[2]));
    }
}
template <bool PACKED>
PIXO_DEV void quant_row8(const float *x, const QPair *r, qtab_t q, float scale, uint32_t out[4])
{
    quant_row4<PACKED>(x, r, q, scale, out);
    PIXO_PIN(out[0]); PIXO_PIN(out[1]);
    quant_row4<PACKED>(x + 4, r + 4, q + 4, scale, out + 2);
}

// Block kinds (wave-uniform): quantiser table, DC shift of the row pass, scale.
//   luminance, chroma 4:4:4   samples b, level shift 128     row DC shift 8*128 = 1024
//   chroma 4:2:0 (U16)        complemented 2x2 sums S' = 1020 - S, S = 4*mean; transform at -4x scale:
//                             mean - 128 = (S - 512)/4 = -(S' - 508)/4      row DC shift 8*508 = 4064, scale -1/4
// `src` points at this lane's first planar row; rows are `pitch` bytes apart and are read from
// LDS as the row pass needs them (2-4 registers), not staged in 16-32 registers.
// The two passes exist in two forms with the SAME arithmetic (aan8_pair on float or on pairs):
//   scalar   one row / one column at a time, plain v_add / v_sub / v_mul_f32
//   packed   two rows / two neighbouring columns at a time, v_pk_add_f32 / v_pk_mul_f32 (round 4: columns, round 5: rows)
// Which one a LAUNCH uses is decided by measurement (profiles/r05_ab_scalar_vs_packed.txt, same-process A/B of library variants,
// 21 rounds of 200 launches): a launch of ONE generation of workgroups — at most 2048, the metric's 4096x4096 4:2:0 image —
// ends with one or two wavefronts per SIMD, which cannot hide the packed instructions' latency: it is 0.5 us FASTER with the
// scalar passes and the plain quantiser although they issue 20 % more vector instructions (17.86 us = 1.0006 x the plain copy
// of the same bytes, where the round-4 kernel took 18.36 and packed rows + columns 18.6-18.9).  Launches of several generations
// are issue-bound and want the packed forms: 4096x4096 4:4:4 30.0 against 31.0 us, the 64 x 1080p batch 138.9 against 143.8.
// So the kernels exist in both forms and the launcher picks by the number of workgroups (jpeg_kernels.hip, packed_launch()).
// Scheduling fences of the scalar form: none in the column pass (the compiler interleaves the eight columns; c8 in the profile),
// the row pass pinned row by row (r1), the quantiser's rows unfenced (q0): together another 0.3 us.
template <bool U16, bool PACKED>
PIXO_DEV void block_rows(const uint8_t *src, int pitch, float dc_shift, float *v)
{
    // all eight planar rows are fetched first (16 or 32 registers that are free this early:
    // v[] fills up only as the rows are transformed) so that the LDS latency is paid once
    u32x4 raw16[U16 ? 8 : 1];
    u32x2 raw8[U16 ? 1 : 8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        if (U16) raw16[r] = *(const u32x4 *)(src + r * pitch);
        else raw8[r] = *(const u32x2 *)(src + r * pitch);
    }
    PIXO_SCHED_FENCE();
    if (!PACKED) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            float d[8], o[8];
            if (U16) {
                const u32x4 w = raw16[r];
                d[0] = (float)(w.x & 0xFFFF); d[1] = (float)(w.x >> 16); d[2] = (float)(w.y & 0xFFFF); d[3] = (float)(w.y >> 16);
                d[4] = (float)(w.z & 0xFFFF); d[5] = (float)(w.z >> 16); d[6] = (float)(w.w & 0xFFFF); d[7] = (float)(w.w >> 16);
            } else {
                const uint32_t lo = raw8[r].x, hi = raw8[r].y;
                d[0] = (float)(lo & 0xFF); d[1] = (float)((lo >> 8) & 0xFF); d[2] = (float)((lo >> 16) & 0xFF); d[3] = (float)(lo >> 24);
                d[4] = (float)(hi & 0xFF); d[5] = (float)((hi >> 8) & 0xFF); d[6] = (float)((hi >> 16) & 0xFF); d[7] = (float)(hi >> 24);
            }
            // (keeps LLVM from folding the first butterflies into integer SDWA adds + conversions: twice as many half-rate instructions)
#pragma unroll
            for (int i = 0; i < 8; i++) PIXO_PIN(d[i]);
            aan8_pair<true, float>(dc_shift, d, o);
            float *lo = &v[r * 8];
            lo[0] = o[0] * PIXO_S0; lo[1] = o[1] * PIXO_S1; lo[2] = o[2] * PIXO_S2; lo[3] = o[3] * PIXO_S3;
            lo[4] = o[4] * PIXO_S4; lo[5] = o[5] * PIXO_S5; lo[6] = o[6] * PIXO_S6; lo[7] = o[7] * PIXO_S7;
            // finish the row (scale multiplications included) before the next one starts: left free, the compiler batches
            // all 64 scale multiplications after row 7 into fresh registers
#pragma unroll
            for (int i = 0; i < 8; i++) PIXO_PIN(v[r * 8 + i]);
            PIXO_SCHED_FENCE();
        }
        return;
    }
    const pixo_cf2 shift2 = {dc_shift, dc_shift};
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        pixo_cf2 d[8], o[8];
        if (U16) rows2_from_u16(raw16[r], raw16[r + 1], d);
        else rows2_from_bytes(raw8[r], raw8[r + 1], d);
        aan8_pair<true>(shift2, d, o);
        float *lo = &v[r * 8], *hi = &v[r * 8 + 8];
        lo[0] = o[0].x * PIXO_S0; hi[0] = o[0].y * PIXO_S0; lo[1] = o[1].x * PIXO_S1; hi[1] = o[1].y * PIXO_S1;
        lo[2] = o[2].x * PIXO_S2; hi[2] = o[2].y * PIXO_S2; lo[3] = o[3].x * PIXO_S3; hi[3] = o[3].y * PIXO_S3;
        lo[4] = o[4].x * PIXO_S4; hi[4] = o[4].y * PIXO_S4; lo[5] = o[5].x * PIXO_S5; hi[5] = o[5].y * PIXO_S5;
        lo[6] = o[6].x * PIXO_S6; hi[6] = o[6].y * PIXO_S6; lo[7] = o[7].x * PIXO_S7; hi[7] = o[7].y * PIXO_S7;
#pragma unroll
        for (int i = 0; i < 16; i++) PIXO_PIN(v[r * 8 + i]);
        PIXO_SCHED_FENCE();
    }
}

template <bool PACKED> PIXO_DEV void block_cols(float *v)
{
    if (!PACKED) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            float d[8], o[8];
#pragma unroll
            for (int r = 0; r < 8; r++) d[r] = v[8 * r + c];
            aan8_pair<false, float>(0.0f, d, o);
            v[c] = o[0] * PIXO_S0; v[8 + c] = o[1] * PIXO_S1; v[16 + c] = o[2] * PIXO_S2; v[24 + c] = o[3] * PIXO_S3;
            v[32 + c] = o[4] * PIXO_S4; v[40 + c] = o[5] * PIXO_S5; v[48 + c] = o[6] * PIXO_S6; v[56 + c] = o[7] * PIXO_S7;
        }
    } else {
        // pairs of neighbouring columns: v[8 r + 2 k], v[8 r + 2 k + 1] as one aligned register pair
#pragma unroll
        for (int k = 0; k < 4; k++) {
            pixo_cf2 d[8], o[8];
#pragma unroll
            for (int r = 0; r < 8; r++) { d[r] = (pixo_cf2){v[8 * r + 2 * k], v[8 * r + 2 * k + 1]}; PIXO_PIN2(d[r]); }
            aan8_pair<false>((pixo_cf2){0.0f, 0.0f}, d, o);
            o[0] = o[0] * PIXO_S0; o[1] = o[1] * PIXO_S1; o[2] = o[2] * PIXO_S2; o[3] = o[3] * PIXO_S3;
            o[4] = o[4] * PIXO_S4; o[5] = o[5] * PIXO_S5; o[6] = o[6] * PIXO_S6; o[7] = o[7] * PIXO_S7;
#pragma unroll
            for (int r = 0; r < 8; r++) { PIXO_PIN2(o[r]); v[8 * r + 2 * k] = o[r].x; v[8 * r + 2 * k + 1] = o[r].y; }
            PIXO_SCHED_FENCE();
        }
    }
#pragma unroll
    for (int i = 0; i < 64; i++) PIXO_PIN(v[i]);
}

// What the block of lane `lane` of consumer wave `wave` is, and where its planar rows start
// (everything wave-uniform except src).
struct BlockDesc {
    const uint8_t *src;
    int pitch;
    int rcp_off, q_off; // offsets into the quantiser table block ((rlo, rhi) pairs from rcp_off)
    float dc_shift, scale;
    bool u16;
};

template <int MODE> PIXO_DEV BlockDesc block_desc(int wave, int lane, const uint8_t *planar)
{
    BlockDesc d;
    d.u16 = false; d.pitch = kPitch; d.rcp_off = 0; d.q_off = 128; d.dc_shift = 1024.0f; d.scale = 1.0f;
    d.src = planar + wave * 4224 + lane * 8;
    if (MODE == M420) {
        if (wave < 2) {
            const int ml = lane >> 2, s = lane & 3;
            d.src = planar + wave * 4352 + ((s >> 1) * 8) * kPitchHalf + ml * 16 + (s & 1) * 8;
            d.pitch = kPitchHalf;
        } else {
            d.src = planar + 8704 + (lane >> 5) * 4096 + (lane & 31) * 16;
            d.pitch = 512; d.u16 = true; d.rcp_off = 384; d.q_off = 192; d.dc_shift = kChromaSumShift; d.scale = kChromaSumScale;
        }
    } else if (MODE == M444) {
        if (wave >= 1) { d.rcp_off = 256; d.q_off = 192; }
    }
    return d;
}

// Consumer step 1: planar rows -> registers, row pass.
template <int MODE, bool PACKED> PIXO_DEV void consumer_rows(int wave, int lane, const uint8_t *planar, float *v)
{
    const BlockDesc d = block_desc<MODE>(wave, lane, planar);
    if (MODE == M420 && d.u16) block_rows<true, PACKED>(d.src, 512, kChromaSumShift, v);
    else block_rows<false, PACKED>(d.src, d.pitch, d.dc_shift, v);
}

// Consumer step 2: column pass (registers only).
template <bool PACKED> PIXO_DEV void consumer_cols(float *v) { block_cols<PACKED>(v); }

// ---------------------------------------------------------------------------------
// Whole-block write-out.  Two 64-byte halves of a 128-byte block stored a microsecond apart cost
// HBM 20 % of its throughput (tools/ubench/tile_copy.hip: 22.1 us against 18.0 us for the same
// bytes as whole lines), and the 4 KiB stage of a wavefront holds only 32 whole blocks.
// So the lane quantises its whole block into 32 registers (the 64 floats die as it goes), and the
// wavefront writes out in two rounds: lanes [32 h, 32 h + 32) put their blocks into the stage, all
// 64 lanes read them back as 16-byte chunks — eight consecutive lanes = one block — and every
// store instruction writes 1 KiB of consecutive bytes.
// ---------------------------------------------------------------------------------
// Chunk (block bl in 0..31, row r in 0..7) lives at 128 bl + 16 ((r ^ bl) & 7): the eight lanes a
// ds_write_b128 group serves (consecutive blocks, same row) and the eight lanes of a read-back group
// (same block, rows 0..7) each touch all eight 16-byte slots of a 128-byte bank window.
PIXO_DEV int stage_addr_block(int bl, int r) { return bl * 128 + (((r ^ bl) & 7) << 4); }

// Consumer step 3': quantise the lane's block, row r -> out[4 r .. 4 r + 3] (packed i16 pairs).
// The reciprocals are wave-uniform scalar (SMEM) loads, fetched one row ahead of their use so that
// their latency hides under the previous row's arithmetic.
template <bool PACKED>
PIXO_DEV void block_quant(const float *v, qtab_t rcp, qtab_t q, float scale, uint32_t *out)
{
    QPair r[8], r_n[8]; // (rlo, rhi) of coefficient i at rcp[2 i], rcp[2 i + 1]: one aligned scalar register pair each
#pragma unroll
    for (int c = 0; c < 8; c++) { r[c].lo = rcp[2 * c]; r[c].hi = rcp[2 * c + 1]; }
    PIXO_SCHED_FENCE();
#pragma unroll
    for (int u = 0; u < 8; u++) {
        if (u < 7) {
#pragma unroll
            for (int c = 0; c < 8; c++) { r_n[c].lo = rcp[2 * ((u + 1) * 8 + c)]; r_n[c].hi = rcp[2 * ((u + 1) * 8 + c) + 1]; }
            PIXO_SCHED_FENCE();
        }
        quant_row8<PACKED>(&v[u * 8], r, q + u * 8, scale, &out[u * 4]);
        // the row's four result registers exist from here on (and its eight floats are dead)
        PIXO_PIN(out[u * 4]); PIXO_PIN(out[u * 4 + 1]); PIXO_PIN(out[u * 4 + 2]); PIXO_PIN(out[u * 4 + 3]);
        if (PACKED) PIXO_SCHED_FENCE(); // (the scalar form runs the rows unfenced: profiles/r05_ab_scalar_vs_packed.txt)
#pragma unroll
        for (int c = 0; c < 8; c++) r[c] = r_n[c];
    }
}
template <int MODE, bool PACKED> PIXO_DEV void consumer_quant(int wave, int lane, const float *qt, const float *v, uint32_t *out)
{
    const BlockDesc d = block_desc<MODE>(wave, lane, nullptr);
    const qtab_t tab = as_qtab(qt);
    block_quant<PACKED>(v, tab + uniform_i32(d.rcp_off), tab + uniform_i32(d.q_off), d.scale, out);
}

// Consumer step 4' (round h = 0, 1): the lanes of half h stage their blocks.
PIXO_DEV void consumer_stage_blocks(int lane, int h, const uint32_t *qw, uint8_t *stage_wave)
{
    if ((lane >> 5) != h) return;
    const int bl = lane & 31;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        u32x4 o;
        o.x = qw[4 * r]; o.y = qw[4 * r + 1]; o.z = qw[4 * r + 2]; o.w = qw[4 * r + 3];
        *(u32x4 *)(stage_wave + stage_addr_block(bl, r)) = o;
    }
}
// (every lane, between the staging of a round and its read-back, and again before the next round)
PIXO_DEV void consumer_stage_sync() { PIXO_WAVE_SYNC(); }

// Consumer step 5' (round h): the 32 staged blocks -> HBM.  Lane l of instruction k moves chunk
// 64 k + l = (block 8 k + l / 8, row l % 8) of the round, i.e. bytes [1024 k + 16 l, + 16) behind the
// round's first block: a wave-uniform 64-bit base (scalar registers) + a 32-bit lane offset, so that
// the stores take the saddr + voffset form and no 64-bit vector arithmetic is spent on addresses.
template <int MODE, bool GUARD>
PIXO_DEV void store_blocks_body(const TileCtx &c, uint32_t u0, uint32_t nvalid, uint32_t tile_y, int wave, int lane,
                                int h, const uint8_t *stage_wave)
{
    const uint32_t lane_off = (uint32_t)lane * 16u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int ch = k * 64 + lane, sb = ch >> 3, j = ch & 7, bl = h * 32 + sb; // bl: block of the wave
        const u32x4 w = *(const u32x4 *)(stage_wave + stage_addr_block(sb, j));
        const uint32_t off = (uint32_t)k * 1024u + lane_off;
        if (MODE == M420) {
            const size_t mcu0 = (size_t)tile_y * c.units_x + u0;
            if (wave < 2) {
                uint8_t *base = (uint8_t *)(c.y + (mcu0 * 4 + (size_t)(wave * 64 + h * 32)) * 64);
                if (!GUARD || (uint32_t)(wave * 16 + (bl >> 2)) < nvalid) PIXO_GSTORE(base + off, w);
            } else { // round 0: the 32 Cb blocks, round 1: the 32 Cr blocks
                uint8_t *base = (uint8_t *)((h == 0 ? c.cb : c.cr) + mcu0 * 64);
                if (!GUARD || (uint32_t)sb < nvalid) PIXO_GSTORE(base + off, w);
            }
        } else if (MODE == M444) {
            const size_t blk = ((size_t)tile_y * c.units_x + u0 + (size_t)(h * 32)) * 64;
            // (wave-uniform branches, one store each, inside a lane-divergent guard — kept even for interior
            // tiles: without it the three stores are merged into one store through a select among
            // c.y / c.cb / c.cr, which the compiler implements in scratch memory)
            if ((uint32_t)bl < nvalid) {
                if (wave == 0) PIXO_GSTORE((uint8_t *)(c.y + blk) + off, w);
                else if (wave == 1) PIXO_GSTORE((uint8_t *)(c.cb + blk) + off, w);
                else PIXO_GSTORE((uint8_t *)(c.cr + blk) + off, w);
            }
        } else {
            const uint32_t brow = tile_y * 3 + wave;
            uint8_t *base = (uint8_t *)(c.y + ((size_t)brow * c.units_x + u0 + (size_t)(h * 32)) * 64);
            if (!GUARD || ((uint32_t)bl < nvalid && brow < c.units_y)) PIXO_GSTORE(base + off, w);
        }
    }
}

template <int MODE>
PIXO_DEV void consumer_store_blocks(const TileCtx &c, uint32_t tile_x, uint32_t tile_y, int wave, int lane, int h,
                                    const uint8_t *stage)
{
    typedef Geo<MODE> G;
    const uint32_t u0 = tile_x * G::units_x;
    const uint32_t nvalid = c.units_x - u0 < (uint32_t)G::units_x ? c.units_x - u0 : G::units_x;
    const bool inside = nvalid == (uint32_t)G::units_x && (MODE != MGRAY || tile_y * 3 + wave < c.units_y);
    if (MODE != M444 && inside) store_blocks_body<MODE, false>(c, u0, nvalid, tile_y, wave, lane, h, stage);
    else store_blocks_body<MODE, true>(c, u0, nvalid, tile_y, wave, lane, h, stage);
}

} // namespace pixo_tile
