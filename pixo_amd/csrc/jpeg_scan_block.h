// jpeg_scan_block.h — per-block body of the device entropy stage: one LANE walks one 8x8 block
// of quantised coefficients in zig-zag order exactly like the reference's encode_block
// (src/jpeg/huffman.rs:423-481) and hands every Huffman symbol to a visitor.  Three visitors
// exist: bit LENGTH of the block, PACK the bits at a known bit offset, symbol HISTOGRAM
// (count_block, src/jpeg/mod.rs:826-860).
//
// Shared between the gfx950 kernels (jpeg_entropy.hip) and the CPU emulation harness in
// tests/emu (-DPIXO_EMU): the walk, the magnitude categories, the value bits and the
// MSB-first packing are all plain integer code.
#pragma once
#include <stdint.h>

#if defined(PIXO_EMU)
#define PIXO_SDEV static inline // free functions
#define PIXO_SMEM inline        // member functions
#define PIXO_SHOST static inline // ... that the host side of the library calls as well
#else
#define PIXO_SDEV __device__ __forceinline__
#define PIXO_SMEM __device__ __forceinline__
#define PIXO_SHOST __host__ __device__ inline
#endif

namespace pixo_scan {

// Huffman tables as the kernels see them: one u32 per symbol = (code length << 16) | code,
// [class 0 = luminance, 1 = chrominance][0..11 DC categories, 12..267 AC run/size symbols].
constexpr int kDcSyms = 12, kAcSyms = 256, kClassSyms = kDcSyms + kAcSyms, kTableWords = 2 * kClassSyms;

// zig-zag scan order, quantize.rs:18-22: natural index of the k-th coefficient of the scan.
// Only ever called with a compile-time k (fully unrolled walk), so it folds to a constant.
PIXO_SDEV constexpr int zigzag(int k)
{
    constexpr uint8_t t[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                               12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                               35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                               58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    return t[k];
}

// `category` (huffman.rs:394-401): number of bits of |v|
PIXO_SDEV int magnitude_bits(int v)
{
    const unsigned a = (unsigned)(v < 0 ? -v : v);
    return a == 0 ? 0 : 32 - __builtin_clz(a);
}

// `encode_value` (huffman.rs:404-418): the low `cat` bits of v (v - 1 for negative v)
PIXO_SDEV uint32_t value_bits(int v, int cat) { return (uint32_t)(v < 0 ? v - 1 : v) & ((1u << cat) - 1u); }

// coefficient j (natural order) of a block held as 32 dwords of two i16 each
PIXO_SDEV int coef_of(const uint32_t *w, int j)
{
#if !defined(PIXO_EMU)
    if ((j & 1) == 0) { // (the low half by ONE bit-field extract: left to itself the compiler tests `w << 16` for zero and shifts that back — three
                        //  instructions at a position with a coefficient where extract + compare are two)
        int r;
        asm("v_bfe_i32 %0, %1, 0, 16" : "=v"(r) : "v"(w[j >> 1]));
        return r;
    }
#endif
    return (int)(int16_t)(w[j >> 1] >> (16 * (j & 1)));
}

// Walks one block.  V provides dc(cat, diff), ac(rs, cat, v), zrl(), eob().
template <class V> PIXO_SDEV void walk_block(const uint32_t *w, int prev_dc, V &vis)
{
    const int d0 = coef_of(w, 0);
    const int diff = (int)(int16_t)(d0 - prev_dc); // i16 arithmetic like the reference
    vis.dc(magnitude_bits(diff), diff);
    int run = 0;
#pragma unroll
    for (int k = 1; k < 64; k++) {
        const int v = coef_of(w, zigzag(k));
        if (v == 0) {
            run++;
        } else {
            while (run >= 16) { vis.zrl(); run -= 16; }
            const int cat = magnitude_bits(v);
            vis.ac((run << 4) | cat, cat, v);
            run = 0;
        }
    }
    if (run > 0) vis.eob();
}

// ---- visitor 1: bit length of the block ----------------------------------------------------
struct LengthVisitor {
    const uint32_t *tab; // this class's kClassSyms words
    uint32_t bits;
    PIXO_SMEM void dc(int cat, int) { bits += (tab[cat] >> 16) + cat; }
    PIXO_SMEM void ac(int rs, int cat, int) { bits += (tab[kDcSyms + rs] >> 16) + cat; }
    PIXO_SMEM void zrl() { bits += tab[kDcSyms + 0xF0] >> 16; }
    PIXO_SMEM void eob() { bits += tab[kDcSyms] >> 16; }
    PIXO_SMEM void band_run(int symbol, int nbits, uint32_t) { bits += (tab[kDcSyms + symbol] >> 16) + nbits; }
};

// ---- visitor 2: pack the block's bits at absolute bit offset `pos` of a zeroed stream ------
// The stream is an array of u32 holding the bits MSB first (stream bit k = bit 31 - k % 32 of
// word k / 32; bytes come out big-endian).  Words this block only partly covers (its first and
// last) are OR-ed atomically — neighbouring blocks, packed by other lanes, share them —
// words it covers completely are stored.
struct PackVisitor {
    const uint32_t *tab;
    uint32_t *stream;
    uint64_t acc;     // pending bits, right-aligned
    int pending;      // how many (< 32 between calls)
    uint64_t word;    // index of the word the pending bits belong to
    bool first;       // nothing flushed yet: the next word is the shared first one

    PIXO_SMEM void begin(uint32_t *s, uint64_t pos)
    {
        stream = s; acc = 0; pending = (int)(pos & 31); word = pos >> 5; first = true;
    }
    PIXO_SMEM void or_word(uint64_t i, uint32_t v)
    {
#if defined(PIXO_EMU)
        stream[i] |= v;
#else
        if (v) atomicOr(&stream[i], v);
#endif
    }
    PIXO_SMEM void put(uint32_t v, int n) // n <= 32
    {
        acc = (acc << n) | v;
        pending += n;
        if (pending >= 32) {
            const uint32_t out = (uint32_t)(acc >> (pending - 32));
            if (first) { or_word(word, out); first = false; }
            else stream[word] = out;
            word++;
            pending -= 32;
            acc &= (1ull << pending) - 1ull;
        }
    }
    PIXO_SMEM void finish() // the last, partly filled word
    {
        if (pending > 0) or_word(word, (uint32_t)(acc << (32 - pending)));
    }
    PIXO_SMEM void dc(int cat, int diff)
    {
        const uint32_t t = tab[cat];
        put(((t & 0xFFFF) << cat) | value_bits(diff, cat), (int)(t >> 16) + cat);
    }
    PIXO_SMEM void ac(int rs, int cat, int v)
    {
        const uint32_t t = tab[kDcSyms + rs];
        put(((t & 0xFFFF) << cat) | value_bits(v, cat), (int)(t >> 16) + cat);
    }
    PIXO_SMEM void zrl() { const uint32_t t = tab[kDcSyms + 0xF0]; put(t & 0xFFFF, (int)(t >> 16)); }
    PIXO_SMEM void eob() { const uint32_t t = tab[kDcSyms]; put(t & 0xFFFF, (int)(t >> 16)); }
    PIXO_SMEM void band_run(int symbol, int nbits, uint32_t extra)
    {
        const uint32_t t = tab[kDcSyms + symbol];
        put(t & 0xFFFF, (int)(t >> 16));
        if (nbits) put(extra, nbits);
    }
};

// ---- visitor 3: symbol statistics for optimised tables ---------------------------------------
struct CountVisitor {
    uint32_t *hist; // this class's kClassSyms counters (LDS on the device)
    PIXO_SMEM void bump(int i)
    {
#if defined(PIXO_EMU)
        hist[i]++;
#else
        atomicAdd(&hist[i], 1u);
#endif
    }
    PIXO_SMEM void dc(int cat, int) { bump(cat); }
    PIXO_SMEM void ac(int rs, int, int) { bump(kDcSyms + (rs & 0xFF)); }
    PIXO_SMEM void zrl() { bump(kDcSyms + 0xF0); }
    PIXO_SMEM void eob() { bump(kDcSyms); }
};

// ---- the same walk without data-dependent branches (jpeg_scan_fused.hip) ------------------------------------------
// walk_block above is written like the reference: per coefficient an if/else, a while loop for the 16-zero runs,
// and inside the visitors more ifs.  On a wavefront every one of those is an exec-mask region — save, and, branch,
// restore: ~70 scalar instructions per coefficient position, 4400 per block, and the scalar unit is shared by the
// whole CU: that, not the vector work, bounded the entropy kernels.  The flat form below computes every position's
// contribution with selects; the only branches left are wave-uniform (a position at which no lane of the wavefront
// holds a non-zero coefficient is skipped; 16-zero runs, rare, take a uniform side path).
#if defined(PIXO_EMU)
#define PIXO_BALLOT64(pred) ((uint64_t)((pred) ? 1 : 0))
PIXO_SDEV uint32_t scan_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31)); }
PIXO_SDEV uint32_t scan_sign_bits(int x) // v_ffbh_i32: leading bits equal to the sign bit; -1 for 0 and -1
{
    if (x == 0 || x == -1) return 0xFFFFFFFFu;
    return (uint32_t)__builtin_clz((unsigned)(x < 0 ? ~x : x));
}
#else
#define PIXO_BALLOT64(pred) __builtin_amdgcn_ballot_w64(pred)
PIXO_SDEV uint32_t scan_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
PIXO_SDEV uint32_t scan_sign_bits(int x)
{
    uint32_t r;
    asm("v_ffbh_i32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
#endif
#define PIXO_ANY64(pred) (PIXO_BALLOT64(pred) != 0)

// The tables as the flat walk wants them: per class
//   [0, 16)     DC, slot = (32 - category) & 15 — category 0 -> 0, 1 -> 15, ... 11 -> 5
//   [16, 272)   AC, slot = run << 4 | ((32 - category) & 15), run < 16
//   272, 273    end of block, run of 16 zeros
// and one word per symbol that does the work of several instructions per coefficient:
//   bits 31..16  the code, LEFT-aligned in 16 bits        byte 0  the code's length
//   byte 1       the code's length + 32 (the walk subtracts m = 32 - category: what is left is length + category)
// AC slots of category 0 — where the walk looks for a coefficient that is zero — say "nothing": no code, no bits.
// The index (32 - category) & 15 is what falls out of v_ffbh_i32 without a subtraction.
constexpr int kWalkDc = 16, kWalkAc = 256, kWalkEob = kWalkDc + kWalkAc, kWalkZrl = kWalkEob + 1, kWalkClassWords = kWalkZrl + 1;
constexpr int kWalkWords = 2 * kWalkClassWords, kScanTableUpload = kTableWords + kWalkWords;
constexpr uint32_t kWalkNothing = 32u << 8;
PIXO_SHOST uint32_t walk_word(uint32_t packed) // (length << 16) | code  ->  the form above
{
    const uint32_t len = packed >> 16, code = packed & 0xFFFFu;
    return len ? ((code << (16u - len)) << 16) | ((len + 32u) << 8) | len : kWalkNothing;
}
// word i of the flat walk's tables from the packed tables (kTableWords words); the host builds them once per scan and
// uploads them behind the packed ones: kScanTableUpload words in all
PIXO_SHOST uint32_t walk_table_word(const uint32_t *packed, int i)
{
    const int cls = i / kWalkClassWords, slot = i % kWalkClassWords;
    const uint32_t *t = packed + cls * kClassSyms;
    if (slot == kWalkEob) return walk_word(t[kDcSyms]);
    if (slot == kWalkZrl) return walk_word(t[kDcSyms + 0xF0]);
    const int c4 = slot & 15, cat = c4 ? 32 - (c4 | 16) : 0;
    if (slot < kWalkDc) return cat < kDcSyms ? walk_word(t[cat]) : kWalkNothing;
    const int run = (slot - kWalkDc) >> 4;
    return cat ? walk_word(t[kDcSyms + ((run << 4) | cat)]) : kWalkNothing;
}

// Packing with selects.  `Sink` provides or_word(flush, word, value): OR `value` into word `word` of the destination
// when `flush` — every word of the destination starts out zero, so complete words and the partial first / last word of
// a block are written the same way — or, a sink that owns its words alone, store it whether `flush` or not (the word
// is stored again, more complete, until the packer moves on: the last store wins).
template <class Sink> struct FlatPack {
    Sink sink;
    uint32_t acc;     // pending bits, left-aligned, zeros below
    uint32_t pending; // < 32
    uint32_t word;    // index of the word the pending bits belong to
    PIXO_SMEM void put_left(uint32_t vl, uint32_t n) // 0 <= n <= 27 bits at the top of vl, zeros below (vl = 0 when n = 0)
    {
        const uint32_t merged = acc | (vl >> pending);
        const uint32_t spill = scan_alignbit(vl, 0u, pending); // the bits that did not fit: low word of {vl, 0} >> pending
        const uint32_t total = pending + n;
        const bool flush = total >= 32u;
        sink.or_word(flush, word, merged);
        word += flush ? 1u : 0u;
        pending = total & 31u;
        acc = flush ? spill : merged;
    }
    PIXO_SMEM void finish() { sink.or_word(pending != 0u, word, acc); }
};

// One symbol: table word `t` (above), the coefficient as u = v - (v < 0) — its low `category` bits are the value bits,
// every bit above them equals the sign — and m = 32 - category.
template <class Sink> PIXO_SDEV void put_symbol(FlatPack<Sink> &p, uint32_t t, uint32_t u, uint32_t m)
{
    const uint32_t value_left = u << (m & 31u);                       // the value bits at the top (category 0: u = 0)
    p.put_left((t & 0xFFFF0000u) | (value_left >> (t & 0xFFu)), ((t >> 8) & 0xFFu) - m);
}

// The U-FORM of a block (round 6): every coefficient v replaced by u = v - (v < 0) — what encode_value writes (huffman.rs:404-418: the
// low `category` bits of u are the value bits, every bit above them equals the sign; u = 0 only for v = 0, u is never -1).  Two packed
// 16-bit instructions per PAIR of coefficients in front of the walk instead of two 32-bit ones per coefficient inside it.
PIXO_SDEV uint32_t pair_to_u(uint32_t w)
{
#if defined(PIXO_EMU)
    const int lo = (int16_t)(w & 0xFFFFu), hi = (int16_t)(w >> 16);
    return ((uint32_t)(uint16_t)(int16_t)(lo - (lo < 0))) | ((uint32_t)(uint16_t)(int16_t)(hi - (hi < 0)) << 16);
#else
    typedef short s16pair __attribute__((ext_vector_type(2)));
    const s16pair v = __builtin_bit_cast(s16pair, w);
    return __builtin_bit_cast(uint32_t, (s16pair)(v + (v >> 15)));
#endif
}
PIXO_SDEV void block_to_u(uint32_t *w)
{
#pragma unroll
    for (int i = 0; i < 32; i++) w[i] = pair_to_u(w[i]);
}
// One position of the AC walk, any run length.  u: the coefficient in u-form.
// (The zero run is counted by the CALLER, behind the join of its wave-uniform "no lane holds a coefficient here" skip: a skip side with
// code of its own — run16 += 16; continue — made the packer's state a two-sided merge, which the compiler resolved with four register
// copies per position on BOTH sides; with nothing on the skip side the state is updated in place: 27 -> 23 vector instructions a position.)
template <class Sink> PIXO_SDEV void walk_position_u(int k, int u, uint32_t &run16, uint32_t zrl, const uint32_t *wtab, FlatPack<Sink> &p, uint64_t nz_lanes)
{
    const bool nz = u != 0;
    if (k > 16 && __builtin_expect((PIXO_BALLOT64(run16 >= 256u) & nz_lanes) != 0, 0)) { // rare: up to three ZRL codes in front of the symbol
#pragma unroll
        for (uint32_t i = 0; i < 3; i++) {
            const bool on = nz && (run16 >> 8) > i;
            p.put_left(on ? (zrl & 0xFFFF0000u) : 0u, on ? (zrl & 0xFFu) : 0u);
        }
        run16 = nz ? (run16 & 255u) : run16;
    }
    const uint32_t s = scan_sign_bits(u), m = s < 32u ? s : 32u; // u = 0: m = 32, slot 0 of its run = "nothing"
    const uint32_t slot = (k > 16 ? (run16 & 255u) : run16) | (m & 15u); // (a lane without a coefficient here may be anywhere in a long run)
    put_symbol(p, wtab[kWalkDc + slot], (uint32_t)u, m);
}
// The AC part of a block in u-form (positions 1..63 and the end-of-block code).  `wtab`: this class's kWalkClassWords words.
// (Round 5 also measured the positions in CHUNKS of 2 / 4 / 8 with the chunk's table words fetched from LDS together: the fused
// kernel's device time per 4096x4096 file was the same within 1 us for noise, photo and gradient content — the table look-up's
// latency is not what the walk waits for — profiles/r05_walk_chunks.txt; the simpler form stays.)
template <class Sink> PIXO_SDEV void block_pack_flat_ac_u(const uint32_t *uw, const uint32_t *wtab, FlatPack<Sink> &p)
{
    const uint32_t zrl = wtab[kWalkZrl], eob = wtab[kWalkEob];
    uint32_t run16 = 0; // 16 x the zeros since the last non-zero coefficient
#pragma unroll
    for (int k = 1; k < 64; k++) {
        const int u = coef_of(uw, zigzag(k));
        const uint64_t nz_lanes = PIXO_BALLOT64(u != 0);
        if (nz_lanes) walk_position_u(k, u, run16, zrl, wtab, p, nz_lanes); // (wave-uniform on the device)
        run16 = u != 0 ? 0u : run16 + 16u;
    }
    p.put_left(run16 ? (eob & 0xFFFF0000u) : 0u, run16 ? (eob & 0xFFu) : 0u);
}
template <class Sink> PIXO_SDEV void block_pack_flat_ac(const uint32_t *w, const uint32_t *wtab, FlatPack<Sink> &p)
{ // (the same from the block as the quantiser left it: a copy in u-form first — the caller's block is usually dead behind this)
    uint32_t uw[32];
#pragma unroll
    for (int i = 0; i < 32; i++) uw[i] = pair_to_u(w[i]);
    block_pack_flat_ac_u(uw, wtab, p);
}
// The DC symbol of a block as bits at the top of a word (encode_block's first step, huffman.rs:430-437): `left` holds the
// Huffman code followed by the value bits, `len` <= 27 of them.  For a walk whose DC predictor arrives late (the fused
// pixel -> bit stream kernel): the AC part is coded first, from bit 0, and this is placed in front of it afterwards.
struct DcBits { uint32_t left, len; };
PIXO_SDEV DcBits dc_symbol_bits(int dc, int prev_dc, const uint32_t *wtab)
{
    const int diff = (int)(int16_t)(dc - prev_dc); // i16 arithmetic like the reference
    const int u = diff + (diff >> 31);
    const uint32_t s = scan_sign_bits(u), m = s < 32u ? s : 32u;
    const uint32_t t = wtab[m & 15u];
    const uint32_t value_left = (uint32_t)u << (m & 31u);
    DcBits b;
    b.left = (t & 0xFFFF0000u) | (value_left >> (t & 0xFFu));
    b.len = ((t >> 8) & 0xFFu) - m;
    return b;
}
// A whole block: the DC symbol, then the AC part.  (_u: the block in u-form — its DC is u0 - (u0 >> 31) again)
template <class Sink> PIXO_SDEV void block_pack_flat_u(const uint32_t *uw, int prev_dc, const uint32_t *wtab, FlatPack<Sink> &p)
{
    {
        const int u0 = coef_of(uw, 0), dc = u0 - (u0 >> 31);
        const int diff = (int)(int16_t)(dc - prev_dc); // i16 arithmetic like the reference
        const int u = diff + (diff >> 31);
        const uint32_t s = scan_sign_bits(u), m = s < 32u ? s : 32u;
        put_symbol(p, wtab[m & 15u], (uint32_t)u, m);
    }
    block_pack_flat_ac_u(uw, wtab, p);
}
template <class Sink> PIXO_SDEV void block_pack_flat(const uint32_t *w, int prev_dc, const uint32_t *wtab, FlatPack<Sink> &p)
{
    uint32_t uw[32];
#pragma unroll
    for (int i = 0; i < 32; i++) uw[i] = pair_to_u(w[i]);
    block_pack_flat_u(uw, prev_dc, wtab, p);
}

// Symbol statistics with the same walk (count_block, jpeg/mod.rs:826-860): `Bump` provides bump(slot, on, amount) —
// add `amount` to the counter of the walk-table slot `slot` of the block's class when `on`.
// (count_dc = false: the DC symbol is somebody else's to count — the fused statistics kernel's first block of a tile, whose predictor
// lives in another workgroup)
template <class Bump> PIXO_SDEV void block_count_flat(const uint32_t *w, int prev_dc, Bump &h, bool count_dc = true)
{
    {
        const int diff = (int)(int16_t)(coef_of(w, 0) - prev_dc);
        const uint32_t s = scan_sign_bits(diff + (diff >> 31)), m = s < 32u ? s : 32u;
        h.bump(m & 15u, count_dc, 1u);
    }
    uint32_t run16 = 0;
#pragma unroll
    for (int k = 1; k < 64; k++) {
        const int v = coef_of(w, zigzag(k));
        const uint64_t nz_lanes = PIXO_BALLOT64(v != 0);
        const bool nz = v != 0;
        if (nz_lanes) { // (wave-uniform on the device; nothing on the other side: walk_position_u)
            if (k > 16 && __builtin_expect((PIXO_BALLOT64(run16 >= 256u) & nz_lanes) != 0, 0)) {
                h.bump((uint32_t)kWalkZrl, nz && run16 >= 256u, run16 >> 8);
                run16 = nz ? (run16 & 255u) : run16;
            }
            const uint32_t s = scan_sign_bits(v + (v >> 31)), m = s < 32u ? s : 32u;
            h.bump((uint32_t)kWalkDc + ((k > 16 ? (run16 & 255u) : run16) | (m & 15u)), nz, 1u);
        }
        run16 = nz ? 0u : run16 + 16u;
    }
    h.bump((uint32_t)kWalkEob, run16 != 0u, 1u);
}
// where the counter of walk-table slot `slot` (of one class) belongs among the class's kClassSyms symbols; -1: no symbol
PIXO_SHOST int walk_slot_symbol(int slot)
{
    if (slot == kWalkEob) return kDcSyms;
    if (slot == kWalkZrl) return kDcSyms + 0xF0;
    const int c4 = slot & 15, cat = c4 ? 32 - (c4 | 16) : 0;
    if (slot < kWalkDc) return cat < kDcSyms ? cat : -1;
    return cat ? kDcSyms + ((((slot - kWalkDc) >> 4) << 4) | cat) : -1;
}

// ---- progressive scans (simple_progressive_script, progressive.rs:98-110) ------------------------
// Seven single-component scans over the tuple, blocks in STORAGE order (jpeg/mod.rs:1286, :1350):
//   0 DC Y   1 DC Cb   2 DC Cr   3 AC Y 1..10   4 AC Y 11..63   5 AC Cb 1..63   6 AC Cr 1..63
// One lane owns one (scan, block) pair — a "virtual block"; first[i] is the virtual index where scan
// i starts (first[7] = their total).  Tables are packed like the baseline ones, except that a symbol
// the table lacks holds the reference's fallback code (0, 4 bits), progressive.rs:363-381.
struct ProgLayout { uint64_t first[8]; };
PIXO_SDEV int prog_scan_of(const ProgLayout &l, uint64_t v)
{
    int i = 0;
#pragma unroll
    for (int k = 1; k < 7; k++) i += v >= l.first[k] ? 1 : 0;
    return i;
}
PIXO_SDEV int prog_comp(int scan) { return scan < 3 ? scan : (scan <= 4 ? 0 : scan - 4); }
PIXO_SDEV int prog_band(int scan) { return scan == 3 ? 0 : (scan == 4 ? 1 : 2); } // 0: 1..10, 1: 11..63, 2: 1..63

// What an AC scan needs to know about a block before any bit is placed: bit 0 = the band holds a
// non-zero coefficient, bit 1 = its last non-zero lies before the band's end (an end-of-band follows).
template <int SS, int SE> PIXO_SDEV uint32_t band_flags(const uint32_t *w)
{
    int last = -1;
#pragma unroll
    for (int k = SS; k <= SE; k++)
        if (coef_of(w, zigzag(k)) != 0) last = k;
    return last < 0 ? 0u : (1u | (last < SE ? 2u : 0u));
}

// One block of a first AC scan over zig-zag [SS, SE] (progressive.rs:141-210 with al = 0), band not
// empty.  V provides ac(rs, cat, v), zrl().  Returns nothing: the caller knows `last < SE` from band_flags.
template <int SS, int SE, class V> PIXO_SDEV void walk_band(const uint32_t *w, V &vis)
{
    int last = SS;
#pragma unroll
    for (int k = SS; k <= SE; k++)
        if (coef_of(w, zigzag(k)) != 0) last = k;
    int run = 0;
#pragma unroll
    for (int k = SS; k <= SE; k++) {
        const int v = coef_of(w, zigzag(k));
        if (k <= last) {
            if (v == 0) {
                run++;
            } else {
                while (run >= 16) { vis.zrl(); run -= 16; }
                const int cat = magnitude_bits(v);
                vis.ac((run << 4) | cat, cat, v);
                run = 0;
            }
        }
    }
}

// The end-of-band run counter (progressive.rs:156-162, :313-345) seen from one block: `before` =
// the counter when the block is reached = (1 if the previous non-empty block of the scan ended
// before the band's end) + the empty blocks since then, reduced by the flushes at 32767.
constexpr uint32_t kMaxBandRun = 0x7FFF;
template <class V> PIXO_SDEV void emit_band_run(uint32_t run, V &vis)
{ // symbol = floor(log2 run) << 4, then that many low bits of the run
    const int nbits = 31 - __builtin_clz(run);
    vis.band_run(nbits << 4, nbits, run - (1u << nbits));
}

// The run counter on entry to virtual block v of an AC scan.  rank = exclusive prefix count of
// non-empty bands over the virtual order, by_rank[r] = the r-th non-empty virtual block.
PIXO_SDEV uint32_t band_run_before(uint64_t v, uint64_t scan_first, uint64_t rank_v, uint64_t rank_first,
                                   const uint32_t *by_rank, const uint32_t *flags)
{
    const uint64_t r = rank_v - rank_first; // non-empty blocks of this scan in front of v
    if (r == 0) return (uint32_t)((v - scan_first) % kMaxBandRun);
    const uint64_t prev = by_rank[rank_first + r - 1];
    return (uint32_t)((((flags[prev] >> 1) & 1u) + (v - prev - 1)) % kMaxBandRun);
}

// Everything one virtual block contributes to its scan's bit stream.
template <class V>
PIXO_SDEV void prog_emit(int scan, const uint32_t *w, int prev_dc, uint32_t flags, uint32_t before, bool last_of_scan, V &vis)
{
    if (scan < 3) { // encode_dc_scan, jpeg/mod.rs:1248-1310 (al = 0)
        const int diff = (int)(int16_t)(coef_of(w, 0) - prev_dc);
        vis.dc(magnitude_bits(diff), diff);
        return;
    }
    uint32_t counter;
    if (flags & 1u) {
        if (before) emit_band_run(before, vis); // flush the pending run first (progressive.rs:171-174)
        const int band = prog_band(scan);
        if (band == 0) walk_band<1, 10>(w, vis);
        else if (band == 1) walk_band<11, 63>(w, vis);
        else walk_band<1, 63>(w, vis);
        counter = (flags & 2u) ? 1u : 0u; // "if we ended before se, start an EOB run" (:206-209)
    } else {
        counter = before + 1;
        if (counter == kMaxBandRun) { emit_band_run(counter, vis); counter = 0; } // (:160-163)
    }
    if (last_of_scan && counter) emit_band_run(counter, vis); // jpeg/mod.rs:1361-1364
}

// ---- the same scans in ONE pass (prog_code_kernel, jpeg_scan_fused.hip; round 4) -----------------------------------
// The multi-pass form above needs to know every block's run counter before any bit is placed (flags, ranks, the r-th
// non-empty block ...).  The single-pass form codes a block's OWN symbols first, with the flat walk, from bit 0 of the
// lane's scratch, and places what the run counter adds — an end-of-band run in front of a non-empty block, the flush at
// 32767 inside a run of empty blocks, the flush behind the scan's last block — around them afterwards:
//
//   counter on entry to block b  =  C_b mod 32767,   C_b = sum of t_j over j in [p, b),  p = the last non-empty block
//   before b (the scan's first block if there is none, whose sum then starts at it), t_j = 1 for an empty block and for a
//   non-empty one whose last non-zero coefficient lies before the band's end (progressive.rs:156-162, :206-209).
//
// Inside a wavefront C_b comes from two ballots (bits p .. b-1 of the t mask); a lane with no non-empty block before it
// in its wavefront adds what flows in from the wavefronts / groups before (BandCount::carried).
//
// The flat walk over zig-zag [ss, se] of a FIRST AC scan (progressive.rs:141-210 with al = 0): block_pack_flat without
// the DC symbol and without an end-of-block code; ss / se are wave-uniform run-time values (one instance serves the
// three bands of the script), positions outside the band are skipped by a uniform branch.  *any: the band holds a
// non-zero coefficient; *ends_zero: its last coefficient is zero (the block leaves the run counter at 1 — or, empty,
// adds one to it).
template <class Sink>
PIXO_SDEV void band_pack_flat(const uint32_t *w, int ss, int se, const uint32_t *wtab, FlatPack<Sink> &p, bool *any, bool *ends_zero)
{
    const uint32_t zrl = wtab[kWalkZrl];
    uint32_t run16 = 0;
    bool seen = false;
#pragma unroll
    for (int k = 1; k < 64; k++) {
        if (k < ss || k > se) continue; // (wave-uniform)
        const int v = coef_of(w, zigzag(k));
        const uint64_t nz_lanes = PIXO_BALLOT64(v != 0);
        const bool nz = v != 0;
        if (nz_lanes) { // (wave-uniform on the device; nothing on the other side: walk_position_u)
            seen = seen || nz;
            if (k > 16 && __builtin_expect((PIXO_BALLOT64(run16 >= 256u) & nz_lanes) != 0, 0)) { // up to three ZRL codes in front of the symbol
#pragma unroll
                for (uint32_t i = 0; i < 3; i++) {
                    const bool on = nz && (run16 >> 8) > i;
                    p.put_left(on ? (zrl & 0xFFFF0000u) : 0u, on ? (zrl & 0xFFu) : 0u);
                }
                run16 = nz ? (run16 & 255u) : run16;
            }
            const int u = v + (v >> 31);
            const uint32_t s = scan_sign_bits(u), m = s < 32u ? s : 32u;
            const uint32_t slot = (k > 16 ? (run16 & 255u) : run16) | (m & 15u);
            put_symbol(p, wtab[kWalkDc + slot], (uint32_t)u, m);
        }
        run16 = nz ? 0u : run16 + 16u;
    }
    *any = seen;
    *ends_zero = run16 != 0u;
}
// Round 5: ALL AC scans of a block's component in ONE walk over its 63 positions.  split: the script codes the component in
// two bands, 1..10 and 11..63 (luminance) — the walk then notes where the first band's bits end (*first_bits), hands out its
// flags and starts the second band with a fresh zero run, in the same scratch right behind the first band's bits; !split:
// the one band 1..63 (chrominance; *first_bits = all its bits, the second band's flags are false).
template <class Sink>
PIXO_SDEV void bands_pack_flat(const uint32_t *w, bool split, const uint32_t *wtab, FlatPack<Sink> &p, uint32_t *first_bits, bool *any,
                               bool *ends_zero)
{
    const uint32_t zrl = wtab[kWalkZrl];
    uint32_t run16 = 0;
    bool seen = false;
    any[0] = any[1] = ends_zero[0] = ends_zero[1] = false;
#pragma unroll
    for (int k = 1; k < 64; k++) {
        if (k == 11 && split) { // (uniform) the first band ends here
            any[0] = seen; ends_zero[0] = run16 != 0u;
            *first_bits = p.word * 32u + p.pending;
            seen = false; run16 = 0;
        }
        const int v = coef_of(w, zigzag(k));
        const uint64_t nz_lanes = PIXO_BALLOT64(v != 0);
        const bool nz = v != 0;
        if (nz_lanes) { // (wave-uniform on the device; nothing on the other side: walk_position_u)
            seen = seen || nz;
            if (k > 16 && __builtin_expect((PIXO_BALLOT64(run16 >= 256u) & nz_lanes) != 0, 0)) { // up to three ZRL codes in front of the symbol
#pragma unroll
                for (uint32_t i = 0; i < 3; i++) {
                    const bool on = nz && (run16 >> 8) > i;
                    p.put_left(on ? (zrl & 0xFFFF0000u) : 0u, on ? (zrl & 0xFFu) : 0u);
                }
                run16 = nz ? (run16 & 255u) : run16;
            }
            const int u = v + (v >> 31);
            const uint32_t s = scan_sign_bits(u), m = s < 32u ? s : 32u;
            const uint32_t slot = (k > 16 ? (run16 & 255u) : run16) | (m & 15u);
            put_symbol(p, wtab[kWalkDc + slot], (uint32_t)u, m);
        }
        run16 = nz ? 0u : run16 + 16u;
    }
    const int last = split ? 1 : 0;
    any[last] = seen; ends_zero[last] = run16 != 0u;
    if (!split) *first_bits = p.word * 32u + p.pending;
}
// Word j of the `n` bits that begin at bit `from` of a lane's scratch (left-aligned, zero behind the last bit): what the
// gather of a band ORs into the window.  words: the lane's scratch, `limit` words of it hold bits.
PIXO_SDEV uint32_t scratch_bits_word(const uint32_t *words, uint32_t limit, uint32_t from, uint32_t n, uint32_t j)
{
    const uint32_t at = (from >> 5) + j, sh = from & 31u;
    const uint32_t a = at < limit ? words[at] : 0u, b = at + 1 < limit ? words[at + 1] : 0u;
    const uint32_t v = sh ? (a << sh) | (b >> (32u - sh)) : a;
    const uint32_t have = n > 32u * j ? n - 32u * j : 0u; // bits of this word that belong to the run
    return have >= 32u ? v : (have ? v & ~(0xFFFFFFFFu >> have) : 0u);
}
// the DC symbol of a first DC scan (encode_dc_first, progressive.rs:112-133 with al = 0)
template <class Sink> PIXO_SDEV void dc_pack_flat(const uint32_t *w, int prev_dc, const uint32_t *wtab, FlatPack<Sink> &p)
{
    const int diff = (int)(int16_t)(coef_of(w, 0) - prev_dc);
    const int u = diff + (diff >> 31);
    const uint32_t s = scan_sign_bits(u), m = s < 32u ? s : 32u;
    put_symbol(p, wtab[m & 15u], (uint32_t)u, m);
}

// C_b inside one wavefront.  zmask / tmask: ballots of "non-empty" and "t" (lanes beyond the scan's end: neither).
struct BandCount { uint32_t local; bool carried; }; // C_b = local (+ the count flowing into the wavefront when `carried`)
PIXO_SDEV BandCount band_count_in_wave(uint64_t zmask, uint64_t tmask, int lane)
{
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const uint64_t pz = zmask & below;
    BandCount c;
    if (pz) {
        const int p = 63 - __builtin_clzll(pz);
        c.local = (uint32_t)__builtin_popcountll(tmask & below & ~((1ull << p) - 1ull)); // t of lanes p .. lane - 1
        c.carried = false;
    } else {
        c.local = (uint32_t)__builtin_popcountll(tmask & below);
        c.carried = true;
    }
    return c;
}
// what a wavefront hands on: does it hold a non-empty block, and the count behind its last one (all of its t when it holds none)
PIXO_SDEV void band_wave_summary(uint64_t zmask, uint64_t tmask, bool *any, uint32_t *tail)
{
    *any = zmask != 0;
    if (zmask) {
        const int p = 63 - __builtin_clzll(zmask);
        *tail = (uint32_t)__builtin_popcountll(tmask & ~((1ull << p) - 1ull));
    } else {
        *tail = (uint32_t)__builtin_popcountll(tmask);
    }
}
// One end-of-band-run symbol (flush_eob_run, progressive.rs:313-345) as bits at the top of a word: `eob_syms` are the
// packed table words ((length << 16) | code) of the class's symbols 0x00, 0x10 ... 0xE0.  run in 1..32767.
struct BandBits { uint32_t left; uint32_t len; }; // len <= 30
PIXO_SDEV BandBits band_run_bits(uint32_t run, const uint32_t *eob_syms)
{
    const uint32_t nbits = 31u - (uint32_t)__builtin_clz(run);
    const uint32_t t = eob_syms[nbits], clen = t >> 16;
    const uint32_t v = ((t & 0xFFFFu) << nbits) | (run - (1u << nbits));
    BandBits b;
    b.len = clen + nbits;
    b.left = v << (32u - b.len);
    return b;
}
// What the run counter makes a lane emit in front of (`pre`) and behind (`post`) its own symbols.  C: C_b above.
struct BandEdge { BandBits pre, post; };
PIXO_SDEV BandEdge band_edge(uint32_t C, bool live, bool nonempty, bool ends_zero, bool last_of_scan, const uint32_t *eob_syms)
{ // (C < 2^32: a scan has at most 65535 x 65535 / 64 blocks)
    BandEdge e;
    e.pre.left = e.pre.len = e.post.left = e.post.len = 0;
    if (!live) return e;
    const uint32_t before = C % kMaxBandRun;
    uint32_t counter;
    if (nonempty) {
        if (before) e.pre = band_run_bits(before, eob_syms); // flush the pending run first (progressive.rs:171-174)
        counter = ends_zero ? 1u : 0u;                          // (:206-209)
    } else {
        counter = before + 1;
        if (counter == kMaxBandRun) { e.pre = band_run_bits(counter, eob_syms); counter = 0; } // (:160-163)
    }
    if (last_of_scan && counter) e.post = band_run_bits(counter, eob_syms); // jpeg/mod.rs:1361-1364
    return e;
}

// Scan-order position -> (component, block index inside that component's array).
// mode 0 gray: Y;  1 4:4:4: Y Cb Cr per block;  2 4:2:0: Y0 Y1 Y2 Y3 Cb Cr per MCU
// (encode_scan, jpeg/mod.rs:1491-1544).  comp: 0 Y, 1 Cb, 2 Cr.
struct BlockRef { int comp; uint64_t index; };
PIXO_SDEV BlockRef block_of(int mode, uint64_t s)
{
    BlockRef r;
    if (mode == 0) { r.comp = 0; r.index = s; }
    else if (mode == 1) { const uint64_t m = s / 3; const int k = (int)(s - m * 3); r.comp = k; r.index = m; }
    else {
        const uint64_t m = s / 6; const int k = (int)(s - m * 6);
        if (k < 4) { r.comp = 0; r.index = m * 4 + k; } else { r.comp = k - 3; r.index = m; }
    }
    return r;
}

} // namespace pixo_scan
