// png_filter_math.h — the per-group arithmetic of the PNG row-filter kernel (png_filter.hip): SWAR subtract and
// average, the Paeth predictor in packed 16-bit lanes, the score, the neighbour alignment, the reference's
// decision sequences and the checksum terms.  Compiled for gfx950 by png_filter.hip and for the HOST by
// tests/emu (-DPIXO_EMU: three AMD builtins replaced by plain C), where it runs against the oracle without a
// GPU.  Loads, the workgroup's reductions and the write-out stay in png_filter.hip.
#pragma once
#include <stdint.h>

#if defined(PIXO_EMU)
#define PIXO_PDEV static inline
#define PIXO_POPAQUE(x) ((void)0)
static inline uint32_t pixo_sad_u8(uint32_t a, uint32_t b, uint32_t acc)
{
    for (int i = 0; i < 4; i++) { const int x = (a >> (8 * i)) & 255, y = (b >> (8 * i)) & 255; acc += (uint32_t)(x > y ? x - y : y - x); }
    return acc;
}
static inline uint32_t pixo_lerp_u8(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) r |= ((((a >> (8 * i)) & 255) + ((b >> (8 * i)) & 255) + ((c >> (8 * i)) & 1)) >> 1) << (8 * i);
    return r;
}
static inline uint32_t pixo_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{ // v_perm_b32: selector byte 0..3 = that byte of s1, 4..7 = of s0, 12 = 0x00
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t k = (sel >> (8 * i)) & 255u;
        const uint32_t b = k < 4 ? (s1 >> (8 * k)) & 255u : (k < 8 ? (s0 >> (8 * (k - 4))) & 255u : 0u);
        r |= b << (8 * i);
    }
    return r;
}
static inline uint32_t pixo_alignbyte(uint32_t hi, uint32_t lo, int sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * sh)); }
static inline uint32_t pixo_udot4(uint32_t a, uint32_t b)
{
    uint32_t s = 0;
    for (int i = 0; i < 4; i++) s += ((a >> (8 * i)) & 255) * ((b >> (8 * i)) & 255);
    return s;
}
#else
#define PIXO_PDEV __device__ __forceinline__
// (opaque to the optimiser: see paeth2)
#define PIXO_POPAQUE(x) asm volatile("" : "+v"(x))
#define pixo_sad_u8(a, b, acc) __builtin_amdgcn_sad_u8((a), (b), (acc))
#define pixo_alignbyte(hi, lo, sh) __builtin_amdgcn_alignbyte((hi), (lo), (sh))
#define pixo_udot4(a, b) __builtin_amdgcn_udot4((a), (b), 0u, false)
#define pixo_lerp_u8(a, b, c) __builtin_amdgcn_lerp((a), (b), (c))
#define pixo_perm(s0, s1, sel) __builtin_amdgcn_perm((s0), (s1), (sel))
#endif
#if defined(PIXO_EMU)
#define pixo_udot4_acc(a, b, acc) (pixo_udot4((a), (b)) + (acc))
static inline uint32_t pixo_mad_u24(uint32_t a, uint32_t b, uint32_t c) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu) + c; }
#else
#define pixo_udot4_acc(a, b, acc) __builtin_amdgcn_udot4((a), (b), (acc), false)
__device__ __forceinline__ uint32_t pixo_mad_u24(uint32_t a, uint32_t b, uint32_t c)
{ // (spelled out: the compiler splits the product and the sum and re-associates the sum with its neighbours)
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#endif

namespace pixo_png {

// FilterStrategy in the reference's declaration order (src/png/mod.rs:345-364)
enum { S_NONE = 0, S_SUB, S_UP, S_AVERAGE, S_PAETH, S_MINSUM, S_ADAPTIVE, S_ADAPTIVE_FAST, S_BIGRAMS };

constexpr uint32_t kH = 0x80808080u;

typedef short s16x2 __attribute__((ext_vector_type(2)));

PIXO_PDEV uint32_t sub4(uint32_t a, uint32_t b)
{ // per-byte a - b (mod 256), no borrow between bytes
    return ((a | kH) - (b & ~kH)) ^ ((a ^ ~b) & kH);
}
PIXO_PDEV uint32_t sub4_biased(uint32_t a, uint32_t b)
{ // (a - b) ^ 0x80 per byte: what the score takes the distance to 0x80 of — the same six operations as sub4, the score's xor saved
    return ((a | kH) - (b & ~kH)) ^ ((a ^ b) & kH);
}
PIXO_PDEV uint32_t avg4(uint32_t a, uint32_t b)
{ // per-byte floor((a + b) / 2)  (fallback.rs:127: u16 sum, / 2): v_lerp_u8 is ((a + b + carry-in bit) >> 1) per byte
    return pixo_lerp_u8(a, b, 0u);
}
PIXO_PDEV s16x2 as_s(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
PIXO_PDEV uint32_t as_u(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
PIXO_PDEV uint32_t gt_mask(s16x2 x, s16x2 y)
{ // per 16-bit lane: 0xFFFF where x > y  (two packed ops: subtract, arithmetic shift)
    return as_u((y - x) >> 15);
}
// Paeth predictor (fallback.rs:144-159) on two bytes held in the low bytes of 16-bit lanes.
// The reference: p = a + b - c, pa = |p - a|, pb = |p - b|, pc = |p - c|; a if pa <= pb and pa <= pc, else b if pb <= pc,
// else c.  Restated without absolute values (round 3; the 2^24 triples are enumerated in tests/test_emu_png.py and on the
// GPU): with lo = min(a, b), hi = max(a, b) the answer only depends on where c lies —
//     c >= hi: p <= lo, the nearest of the three is lo;     c <= lo: hi;
//     lo < c < hi: p = lo + hi - c lies between them at distances (hi - c) from lo, (c - lo) from hi and |(hi - c) - (c - lo)|
//     from c:  lo if 2 (hi - c) <= c - lo,  hi if 2 (c - lo) <= hi - c,  else c   (ties go to a, b before c: "<=")
// and the outer cases satisfy the same two inequalities, so for ALL c, with x = 3 c - (a + b):
//     predictor = lo if x >= hi,  else hi if x <= lo,  else c.
// The distances needed 17 packed 16-bit operations per two bytes (half rate on this chip: 34 issue slots).  Here: ONE
// packed minimum and two packed arithmetic shifts (6 slots); everything else is plain 32-bit arithmetic on the two lanes
// at once (full rate, 10 slots), which is exact because no lane ever borrows from its neighbour — a + b <= 510 and
// 3 c <= 765 fit a lane, and x is carried with a bias of 0x8000 per lane, so that the lane's bit 15 IS the sign of the
// difference under test.  The kernel is VALU bound and Paeth was 60 % of it.
typedef unsigned short pu16x2 __attribute__((ext_vector_type(2)));
PIXO_PDEV pu16x2 as_us(uint32_t v) { return __builtin_bit_cast(pu16x2, v); }
PIXO_PDEV uint32_t paeth2(uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t lo = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(as_us(a), as_us(b)));
    const uint32_t s = a + b, hi = s - lo;
    const uint32_t xk = pixo_mad_u24(c, 3u, 0x80008000u - s); // lanes: 0x8000 + x, x in [-510, 765]  (0x8000 - s > 0: no borrow)
    uint32_t is_lo = as_u(as_s(xk - hi) >> 15);                // ones where x - hi >= 0 (lanes 0x8000 +- 1020: no borrow)
    uint32_t not_hi = as_u(as_s(xk - lo - 0x00010001u) >> 15); // ones where x - lo - 1 >= 0, i.e. NOT x <= lo
    // (opaque: otherwise the masks are turned back into 16-bit compares + SDWA selects + a permute)
    PIXO_POPAQUE(is_lo); PIXO_POPAQUE(not_hi);
    const uint32_t hc = (c & not_hi) | (hi & ~not_hi);
    return (lo & is_lo) | (hc & ~is_lo);
}
PIXO_PDEV uint32_t paeth4(uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t M = 0x00FF00FFu;
    const uint32_t e = paeth2(a & M, b & M, c & M);
    // (the odd bytes by one byte permute each: a shift by a constant is in the expensive class, profiles/r03_ubench_form_rate.txt)
    const uint32_t o = paeth2(pixo_perm(0u, a, 0x0C030C01u), pixo_perm(0u, b, 0x0C030C01u), pixo_perm(0u, c, 0x0C030C01u));
    return e | (o << 8);
}
PIXO_PDEV uint32_t score4(uint32_t f, uint32_t acc)
{ // sum over the 4 bytes of |byte as i8|: |f - 128 biased| = |(f ^ 0x80) - 0x80|
    return pixo_sad_u8(f ^ kH, kH, acc);
}

enum { F_NONE = 0, F_SUB = 1, F_UP = 2, F_AVG = 3, F_PAETH = 4 };

struct Group { // 16 bytes of the row and their neighbours, as dwords
    uint32_t cur[4], left[4], up[4], ul[4];
    uint32_t valid[4]; // byte mask of the bytes that exist
};

// bytes [4k - BPP, 4k - BPP + 4) of the row as one dword, from X = {dword k-2, dword k-1, dword k}
template <int BPP> PIXO_PDEV uint32_t left_of(const uint32_t *x6, int j)
{ // x6 = {L0, L1, c0, c1, c2, c3}: dword j of the group sits at x6[2 + j]
    constexpr int dummy = 0; (void)dummy;
    const int p = 8 + 4 * j - BPP, idx = p >> 2, sh = p & 3;
    return sh ? pixo_alignbyte(x6[idx + 1], x6[idx], sh) : x6[idx];
}

// The same 16 bytes and their neighbours as loaded: 6 dwords of the row, 6 of the row above.  Rows of at
// most kRegIters * 4 KiB are held like this by the whole workgroup between scoring and write-out.
struct Raw { uint32_t x[6], u[6]; };
// MASK = false: the group lies wholly inside the row (every group but a row's last partial one): no byte masks
template <int BPP, bool MASK> PIXO_PDEV void group_of(const Raw &r, int k0, int n, Group &g)
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        g.cur[j] = r.x[2 + j]; g.up[j] = r.u[2 + j];
        g.left[j] = left_of<BPP>(r.x, j); g.ul[j] = left_of<BPP>(r.u, j);
        const int rem = n - 4 * (k0 + j);
        g.valid[j] = !MASK || rem >= 4 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : (1u << (8 * rem)) - 1u);
    }
}
PIXO_PDEV uint32_t filtered(int f, const Group &g, int j)
{
    switch (f) {
    case F_NONE: return g.cur[j];
    case F_SUB: return sub4(g.cur[j], g.left[j]);
    case F_UP: return sub4(g.cur[j], g.up[j]);
    case F_AVG: return sub4(g.cur[j], avg4(g.left[j], g.up[j]));
    default: return sub4(g.cur[j], paeth4(g.left[j], g.up[j], g.ul[j]));
    }
}

// (FASTMODE: AdaptiveFast never looks at None / Average — a template parameter, not a run-time flag: as a flag the compiler
// computed both candidates and threw them away, AdaptiveFast was no faster than Adaptive)
template <int BPP, bool MASK, bool FASTMODE>
PIXO_PDEV void score_group(const Raw &r, int k0, int n, uint32_t sc[5])
{
    Group g;
    group_of<BPP, MASK>(r, k0, n, g);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!MASK) { // a group wholly inside the row: the subtraction leaves its bytes biased by 0x80, the score needs no xor of its own
            sc[F_SUB] = pixo_sad_u8(sub4_biased(g.cur[j], g.left[j]), kH, sc[F_SUB]);
            sc[F_UP] = pixo_sad_u8(sub4_biased(g.cur[j], g.up[j]), kH, sc[F_UP]);
            sc[F_PAETH] = pixo_sad_u8(sub4_biased(g.cur[j], paeth4(g.left[j], g.up[j], g.ul[j])), kH, sc[F_PAETH]);
            if (!FASTMODE) {
                sc[F_NONE] = score4(g.cur[j], sc[F_NONE]);
                sc[F_AVG] = pixo_sad_u8(sub4_biased(g.cur[j], avg4(g.left[j], g.up[j])), kH, sc[F_AVG]);
            }
            continue;
        }
        const uint32_t m = g.valid[j];
        sc[F_SUB] = score4(filtered(F_SUB, g, j) & m, sc[F_SUB]);
        sc[F_UP] = score4(filtered(F_UP, g, j) & m, sc[F_UP]);
        sc[F_PAETH] = score4(filtered(F_PAETH, g, j) & m, sc[F_PAETH]);
        if (!FASTMODE) {
            sc[F_NONE] = score4(g.cur[j] & m, sc[F_NONE]);
            sc[F_AVG] = score4(filtered(F_AVG, g, j) & m, sc[F_AVG]);
        }
    }
}

// the reference's decision sequences, replayed on the five row scores
template <class T> PIXO_PDEV int decide(int strategy, const T s[5], T n)
{
    if (strategy <= S_PAETH) return strategy; // None, Sub, Up, Average, Paeth
    if (strategy == S_ADAPTIVE_FAST) { // filter.rs:474-527
        const T early = n / 8 + 1;
        int best = F_SUB;
        T bs = s[F_SUB];
        if (bs <= early) return best;
        if (s[F_UP] < bs) { bs = s[F_UP]; best = F_UP; }
        if (bs <= early) return best;
        if (s[F_PAETH] < bs) best = F_PAETH;
        return best;
    }
    // Adaptive / MinSum, filter.rs:302-404: None, Sub, Up, Average, Paeth in this order, a later
    // filter wins only with a strictly smaller score, stop as soon as the best is <= early (or 0)
    const T early = n / 4 + 1;
    int best = F_NONE;
    T bs = s[F_NONE];
    if (bs <= early || bs == 0) return best;
#pragma unroll
    for (int f = F_SUB; f <= F_AVG; f++) {
        if (s[f] < bs) { bs = s[f]; best = f; if (bs == 0 || bs <= early) return best; }
    }
    if (s[F_PAETH] < bs) best = F_PAETH;
    return best;
}


// Checksum terms of one filtered dword whose bytes sit at stream positions p .. p + 3: their sum and
// 3 b0 + 2 b1 + 1 b2 + 0 b3 (the caller adds (weight of the last byte) * sum + ramp to the weighted sum).
PIXO_PDEV void adler_terms(uint32_t v, uint32_t &sum, uint32_t &ramp)
{
    sum = pixo_sad_u8(v, 0u, 0u);
    ramp = pixo_udot4(v, 0x00010203u);
}

// The same for a whole group of four filtered dwords (16 stream bytes): their sum and 15 b0 + 14 b1 + ... + 0 b15 — the
// caller adds (weight of the group's LAST byte) * sum + ramp: one multiplication per group instead of four.
PIXO_PDEV void adler_terms16(const uint32_t v[4], uint32_t &sum, uint32_t &ramp)
{
    sum = pixo_sad_u8(v[3], 0u, pixo_sad_u8(v[2], 0u, pixo_sad_u8(v[1], 0u, pixo_sad_u8(v[0], 0u, 0u))));
    ramp = pixo_udot4_acc(v[0], 0x0C0D0E0Fu, pixo_udot4_acc(v[1], 0x08090A0Bu, pixo_udot4_acc(v[2], 0x04050607u, pixo_udot4(v[3], 0x00010203u))));
}

// Bigrams (score_bigrams, filter.rs:635-649): the pairs (byte p, byte p + 1) that START in filtered dword v and end
// inside the row, as 16-bit keys (any one-to-one pair -> key mapping counts the same number of distinct pairs: the
// little-endian halfword at each byte position).  `next` = the following filtered dword (its first byte closes the
// fourth pair), lim = n - 1 - (position of v's first byte): how many of the four pairs exist.  Returns their number.
PIXO_PDEV int bigram_keys(uint32_t v, uint32_t next, int lim, uint32_t key[4])
{
    key[0] = v & 0xFFFFu; key[1] = (v >> 8) & 0xFFFFu; key[2] = v >> 16; key[3] = (v >> 24) | ((next & 0xFFu) << 8);
    return lim < 0 ? 0 : (lim > 4 ? 4 : lim);
}
// bigrams_filter (filter.rs:406-472): None, Sub, Up, Average, Paeth in this order, a later filter wins only with
// strictly fewer distinct pairs, no early exit
PIXO_PDEV int decide_bigrams(const unsigned long long tot[5])
{
    int f = F_NONE;
#pragma unroll
    for (int c = F_SUB; c <= F_PAETH; c++)
        if (tot[c] < tot[f]) f = c;
    return f;
}

} // namespace pixo_png
