// jpeg_trellis.h — the reference's trellis quantiser (src/jpeg/trellis.rs:67-299, DEFAULT_LAMBDA) for
// one 8x8 block, as run by one LANE of the gfx950 kernel in jpeg_trellis.hip.  Also compiled for the
// host by tests/emu (-DPIXO_EMU) and compared with the oracle there.
//
// A Viterbi search over the 63 AC coefficients in zig-zag order: up to 5 candidate values per
// coefficient (0, floor, round, ceil, one step further out), states keyed by (value, zero run), at
// most 8 survivors per position after a STABLE sort by cost, costs = estimated Huffman bits +
// squared error, all in f32 with one rounding per operation (no FMA: -ffp-contract=off).  Every
// tie rule of the reference is reproduced: candidates in generation order, the first state with an
// equal key is the one replaced (only by a strictly smaller cost), equal costs keep insertion
// order, the first minimum wins at the end.
#pragma once
#include <stdint.h>

#if defined(PIXO_EMU)
#include <math.h>
#define PIXO_TDEV static inline
#else
#define PIXO_TDEV __device__ __forceinline__
#endif

namespace pixo_trellis {

struct State { float cost; int16_t value; uint8_t run; uint8_t parent; };

PIXO_TDEV int16_t to_i16(float v)
{ // Rust `as i16`: saturating
    if (v >= 32767.0f) return 32767;
    if (v <= -32768.0f) return -32768;
    return (int16_t)v;
}
PIXO_TDEV int size_category(int v)
{ // trellis.rs:289-296
    const unsigned a = (unsigned)(v < 0 ? -v : v);
    return a == 0 ? 0 : 32 - __builtin_clz(a);
}
// trellis.rs:210-244
PIXO_TDEV int candidates(float fq, int16_t c[5])
{
    const int16_t r = to_i16(__builtin_roundf(fq)), fl = to_i16(__builtin_floorf(fq)), ce = to_i16(__builtin_ceilf(fq));
    int n = 0;
    c[n++] = 0;
    if (fl != 0) c[n++] = fl;                                   // (only 0 is in the list so far)
    if (r != 0 && r != c[n - 1]) c[n++] = r;                    // r is fl, ce, or new; compared below as well
    bool has;
    has = false; for (int i = 0; i < n; i++) has |= c[i] == ce;
    if (ce != 0 && !has) c[n++] = ce;
    if (__builtin_fabsf(fq) > 1.5f) {
        const int16_t ext = (int16_t)(fq >= 0.0f ? ce + 1 : fl - 1);
        has = false; for (int i = 0; i < n; i++) has |= c[i] == ext;
        if (!has) c[n++] = ext;
    }
    return n;
}
// trellis.rs:246-279: estimated code length of the (run, size) symbol + the value bits
PIXO_TDEV float ac_rate(int value, int run)
{
    const int cat = size_category(value);
    const int rs = (run << 4) | cat;
    float bits;
    switch (rs) {
    case 0x00: bits = 4.0f; break; case 0x01: bits = 2.0f; break; case 0x02: bits = 2.5f; break;
    case 0x03: bits = 3.0f; break; case 0x04: bits = 4.0f; break; case 0x11: bits = 3.0f; break;
    case 0x12: bits = 4.0f; break; case 0x21: bits = 4.0f; break; case 0xF0: bits = 10.0f; break;
    default: bits = 3.0f + (float)(rs >> 4) * 0.5f + (float)(rs & 0x0F) * 0.3f; break;
    }
    return bits + (float)cat;
}

constexpr int kZigzagNat[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// dct, q: natural order.  out: natural order.  trail: 63 x 8 words of back-pointers (value << 8 | parent).
PIXO_TDEV void quantize_block(const float *dct, const float *q, int16_t *out, uint32_t *trail, uint8_t *counts)
{
    for (int i = 0; i < 64; i++) out[i] = 0;
    out[0] = to_i16(__builtin_roundf(dct[0] / q[0])); // DC: plain rounding (trellis.rs:75)
    State cur[8], nxt[16];
    int ncur = 1;
    cur[0].cost = 0.0f; cur[0].value = 0; cur[0].run = 0; cur[0].parent = 0;
    for (int zz = 1; zz < 64; zz++) {
        const int nat = kZigzagNat[zz];
        const float coef = dct[nat], qq = q[nat];
        int16_t cand[5];
        const int nc = candidates(coef / qq, cand);
        int nn = 0;
        for (int pi = 0; pi < ncur; pi++) {
            for (int ci = 0; ci < nc; ci++) {
                const int c = cand[ci];
                const float rec = (float)c * qq, d = coef - rec, dist = d * d;
                float rate;
                int nrun;
                if (c == 0) {
                    const int r = cur[pi].run + 1;
                    if (r >= 16) { rate = 10.0f; nrun = 0; } // a ZRL symbol will be needed (trellis.rs:117-120)
                    else { rate = 0.0f; nrun = r; }
                } else {
                    rate = ac_rate(c, cur[pi].run);
                    nrun = 0;
                }
                const float cost = cur[pi].cost + rate + 1.0f * dist; // lambda = DEFAULT_LAMBDA = 1.0
                int found = -1;
                for (int k = 0; k < nn; k++)
                    if (found < 0 && nxt[k].value == c && nxt[k].run == nrun) found = k;
                if (found < 0) { nxt[nn].cost = cost; nxt[nn].value = (int16_t)c; nxt[nn].run = (uint8_t)nrun; nxt[nn].parent = (uint8_t)pi; nn++; }
                else if (cost < nxt[found].cost) { nxt[found].cost = cost; nxt[found].value = (int16_t)c; nxt[found].run = (uint8_t)nrun; nxt[found].parent = (uint8_t)pi; }
            }
        }
        // stable insertion sort by cost; the 8 cheapest survive (trellis.rs:156-162)
        for (int i = 1; i < nn; i++) {
            const State t = nxt[i];
            int j = i - 1;
            while (j >= 0 && nxt[j].cost > t.cost) { nxt[j + 1] = nxt[j]; j--; }
            nxt[j + 1] = t;
        }
        if (nn > 8) nn = 8;
        for (int i = 0; i < nn; i++) {
            cur[i] = nxt[i];
            trail[(zz - 1) * 8 + i] = ((uint32_t)(uint16_t)nxt[i].value << 8) | nxt[i].parent;
        }
        counts[zz - 1] = (uint8_t)nn;
        ncur = nn;
    }
    for (int i = 0; i < ncur; i++)
        if (cur[i].run > 0) cur[i].cost += 4.0f; // trailing zeros: an EOB will be coded (trellis.rs:172-178)
    int idx = 0;
    for (int i = 1; i < ncur; i++)
        if (cur[i].cost < cur[idx].cost) idx = i; // min_by: the first of equal minima
    for (int zz = 63; zz >= 1; zz--) {
        if (idx < counts[zz - 1]) {
            const uint32_t t = trail[(zz - 1) * 8 + idx];
            out[kZigzagNat[zz]] = (int16_t)(uint16_t)(t >> 8);
            idx = (int)(t & 0xFF);
        }
    }
}

// ---- the same search with every state in registers -------------------------------------------------
// quantize_block above follows the reference's data structures (vectors of states, linear searches);
// run by a GPU lane it lives in scratch memory and is latency-bound.  quantize_block_fast is the same
// function restated on the STRUCTURE of the state set, so that every array has a fixed size, every
// index is static and all loops unroll:
//   * candidates sit in fixed slots by kind (0, floor, round, ceil, one-further) with a valid bit; a
//     slot is valid exactly when the reference would have pushed it, and the slot order is the
//     reference's generation order;
//   * a successor state is keyed by (value, run).  Non-zero values always have run 0, so each non-zero
//     candidate has ONE successor whose cost is the first strict minimum over the parents; a zero
//     successor is keyed by its run alone, and two parents produce the same run only when both have
//     run 0 (all non-zero-valued parents and a (0, 0) parent): those fold into one entry that sits where
//     the first of them would have been inserted;
//   * insertion order therefore is: zero-successor of parent 0, the candidates, the zero-successors of
//     parents 1..7 — twelve slots, of which at most eleven are ever filled (the round and the ceil candidate exclude each
//     other, candidate_kinds3); the reference's stable sort by cost is an 11-input sorting network over 64-bit
//     keys (cost bits, slot) (costs are sums of non-negative terms, so their bit patterns order like
//     the values), the eight smallest survive;
//   * a back-pointer is 6 bits (candidate kind, parent), one byte per survivor: 64 bits per coefficient; values are
//     recomputed from the kind while walking back.
// Env: coef(zz), step(zz) (zig-zag position -> f32), rate_at(4 rs) (ac_rate as a table of rate_value(rs), by byte offset),
// trail_put(pos, u64), trail_get(pos), out(zz, i16).
PIXO_TDEV float rate_bits(int rs)
{ // trellis.rs:246-279 without the value bits; rs = (run << 4) | size, 0..255
    switch (rs) {
    case 0x00: return 4.0f; case 0x01: return 2.0f; case 0x02: return 2.5f;
    case 0x03: return 3.0f; case 0x04: return 4.0f; case 0x11: return 3.0f;
    case 0x12: return 4.0f; case 0x21: return 4.0f; case 0xF0: return 10.0f;
    default: return 3.0f + (float)(rs >> 4) * 0.5f + (float)(rs & 0x0F) * 0.3f;
    }
}
PIXO_TDEV float rate_value(int rs) { return rate_bits(rs) + (float)(rs & 0x0F); } // ac_rate: estimate + value bits (size)
PIXO_TDEV uint32_t f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
PIXO_TDEV float u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
// "No state": as a float a NaN — never smaller than anything, never chosen — and, as the HIGH word of a 64-bit sort key read
// as a double, a finite positive number above every cost (costs are non-negative f32: their bit patterns are at most
// 0x7F800000), so that the sorting network can run on v_min_f64 / v_max_f64 (below).
constexpr uint32_t kNoState = 0x7FEFFFFFu;

// (c ? a : b stays v_cndmask_b32.  The instruction microbenchmarks price it at ~23.5 cycles per wavefront and SIMD
// (profiles/r01_ubench_valu_rates.txt, r03_ubench_form_rate.txt), which would make the ~50 selects per coefficient
// position half of this kernel; replacing 21 of them by a v_mov under a narrowed EXEC mask — v_cmp into a scalar pair,
// three scalar instructions around the move — made the kernel SLOWER, 294 -> 321 us for 4096x4096 (round 4,
// profiles/r04_trellis.txt): inside real code the select does not cost what the isolated loop says, and every write to
// EXEC stalls the vector pipe.)
PIXO_TDEV uint32_t sel_u32(bool c, uint32_t a, uint32_t b) { return c ? a : b; }
PIXO_TDEV int sel_i32(bool c, int a, int b) { return c ? a : b; }

struct Kinds { int v[5]; bool ok[5]; }; // [0] is the zero candidate
PIXO_TDEV Kinds candidate_kinds(float fq)
{
    const int r = to_i16(__builtin_roundf(fq)), fl = to_i16(__builtin_floorf(fq)), ce = to_i16(__builtin_ceilf(fq));
    const int ext = (int16_t)(fq >= 0.0f ? ce + 1 : fl - 1);
    Kinds k;
    k.v[0] = 0; k.ok[0] = true;
    k.v[1] = fl; k.ok[1] = fl != 0;
    k.v[2] = r; k.ok[2] = r != 0 && r != fl;
    k.v[3] = ce; k.ok[3] = ce != 0 && ce != fl && ce != r;
    k.v[4] = ext; k.ok[4] = __builtin_fabsf(fq) > 1.5f && ext != 0 && ext != fl && ext != r && ext != ce;
    return k;
}
// The candidates that can coexist (round 4).  round(fq) is floor(fq) or ceil(fq), so of the kinds 2 (round) and 3 (ceil) at
// most ONE is valid: kind 2 when round != floor (then ceil == round is a duplicate), kind 3 when round == floor and
// ceil != floor.  The search therefore evaluates three non-zero candidates — floor, "the other end", one-further — and the
// middle one remembers which kind it is: its place in the insertion order (sort key) and in the back-pointer is that kind's.
struct Kinds3 { int v[3]; bool ok[3]; uint32_t kind[3]; }; // kinds 1, 2 or 3, 4
PIXO_TDEV Kinds3 candidate_kinds3(float fq)
{
    const int r = to_i16(__builtin_roundf(fq)), fl = to_i16(__builtin_floorf(fq)), ce = to_i16(__builtin_ceilf(fq));
    const int ext = (int16_t)sel_i32(fq >= 0.0f, ce + 1, fl - 1);
    Kinds3 k;
    k.v[0] = fl; k.ok[0] = fl != 0; k.kind[0] = 1;
    const bool is_round = r != fl;
    const int mid = sel_i32(is_round, r, ce);
    k.v[1] = mid; k.ok[1] = mid != 0 && mid != fl; k.kind[1] = 3u - (is_round ? 1u : 0u);
    k.v[2] = ext; k.ok[2] = __builtin_fabsf(fq) > 1.5f && ext != 0 && ext != fl && ext != mid; k.kind[2] = 4;
    return k;
}
// byte 0 of a, b, c, d as one word (a lowest): on the device three v_perm_b32
PIXO_TDEV uint32_t low_bytes(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t ab = __builtin_amdgcn_perm(b, a, 0x0C0C0400u), cd = __builtin_amdgcn_perm(d, c, 0x04000C0Cu);
    return ab | cd;
#else
    return (a & 0xFFu) | ((b & 0xFFu) << 8) | ((c & 0xFFu) << 16) | (d << 24);
#endif
}

// One compare-exchange of the sorting network.  The keys (cost bits, slot, ...) are distinct, non-negative as 64-bit
// integers and — with kNoState as above — finite or denormal positive doubles, which order exactly like the integers:
// on the device a compare-exchange is v_min_f64 + v_max_f64 (two instructions of the double-precision pipe) where the
// 64-bit integer compare and its four selects were about twice that; 39 of them per coefficient position were nearly
// half of the kernel.  (Denormals: the kernel runs with f64 denormals preserved — the HSA default — and min / max return
// an operand unchanged.)
#if defined(__HIP_DEVICE_COMPILE__)
PIXO_TDEV uint64_t min_key(uint64_t a, uint64_t b) { uint64_t r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define PIXO_CE_LO(a, b) { e[a] = min_key(e[a], e[b]); } // (the larger one is not looked at again: checked for wires 8..11 below)
#define PIXO_CE(a, b) { uint64_t lo_, hi_; asm("v_min_f64 %0, %1, %2" : "=v"(lo_) : "v"(e[a]), "v"(e[b])); \
                        asm("v_max_f64 %0, %1, %2" : "=v"(hi_) : "v"(e[a]), "v"(e[b])); e[a] = lo_; e[b] = hi_; }
#else
PIXO_TDEV uint64_t min_key(uint64_t a, uint64_t b) { return a < b ? a : b; }
#define PIXO_CE_LO(a, b) { e[a] = min_key(e[a], e[b]); }
#define PIXO_CE(a, b) { const uint64_t lo_ = e[a] < e[b] ? e[a] : e[b], hi_ = e[a] < e[b] ? e[b] : e[a]; e[a] = lo_; e[b] = hi_; }
#endif

template <class Env> PIXO_TDEV void quantize_block_fast(Env &env)
{
    env.out(0, to_i16(__builtin_roundf(env.coef(0) / env.step(0)))); // DC: plain rounding (trellis.rs:75)
    uint32_t cc[8]; // cost bits of the surviving states, cheapest first; kNoState beyond their number
    uint32_t run6[8]; // the states' zero runs, as run << 6: the byte offset of the run's row in the rate table, and where
                      // the run sits in a sort key's payload
#pragma unroll
    for (int i = 0; i < 8; i++) { cc[i] = kNoState; run6[i] = 0; }
    cc[0] = 0; // cost 0.0
    float coef_next = env.coef(1), step_next = env.step(1); // (fetched a step ahead: on the device a load from HBM)
    for (int zz = 1; zz < 64; zz++) {
        const float coef = coef_next, qq = step_next;
        if (zz < 63) { coef_next = env.coef(zz + 1); step_next = env.step(zz + 1); }
        const Kinds3 k = candidate_kinds3(coef / qq);
        uint64_t e[11];
        // non-zero candidates -> wires 1..3
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float rec = (float)k.v[j] * qq, d = coef - rec, dist = d * d;
            const uint32_t cat4 = (uint32_t)size_category(k.v[j]) << 2;
            // the first strict minimum over the parents = the smallest (cost bits, parent) pair: costs are non-negative (their
            // bit patterns order like their values) and a dead parent's NaN cost is a pattern above every number — one
            // v_min_f64 per parent on the device instead of a compare and two selects
            uint64_t bestk = 0;
#pragma unroll
            for (int pi = 0; pi < 8; pi++) {
                const float rate = env.rate_at(run6[pi] | cat4); // ac_rate(run, size) — byte offset 4 (run << 4 | size)
                const float cost = u2f(cc[pi]) + rate + 1.0f * dist;
                const uint64_t key = ((uint64_t)f2u(cost) << 32) | (uint32_t)pi;
                bestk = pi == 0 ? key : min_key(bestk, key);
            }
            // payload (low word): slot << 28 | run << 6 | kind << 3 | parent — run 0, slot = kind (the reference's generation order)
            const uint32_t lo = (uint32_t)bestk | (k.kind[j] << 28) | (k.kind[j] << 3);
            e[1 + j] = ((uint64_t)sel_u32(k.ok[j], (uint32_t)(bestk >> 32), kNoState) << 32) | lo;
        }
        // zero candidate -> wire 0 (parent 0) and wires 4..10 (parents 1..7).  Successors are keyed by their run; two
        // parents give the same run only when both have run 0 (then the successor is (0, 1)): those fold into ONE entry at
        // the first of them, with the first strict minimum of their costs.  The states are sorted by cost and every member
        // of that group adds the same two terms (+ 0.0, + dist0; f32 addition is monotone), so the first strict minimum IS
        // the first member: the entry is simply that parent's own, the later members vanish.
        const float dist0 = coef * coef; // (coef - 0 * step)^2
        bool seen0 = false;
#pragma unroll
        for (int pi = 0; pi < 8; pi++) {
            const int slot = pi == 0 ? 0 : 4 + pi, wire = pi == 0 ? 0 : 3 + pi;
            const bool alive = cc[pi] != kNoState, run0 = run6[pi] == 0;
            const bool over = run6[pi] == (15u << 6); // a ZRL symbol will be needed (trellis.rs:117-120): run + 1 >= 16
            const float cost = u2f(cc[pi]) + u2f(sel_u32(over, 0x41200000u /* 10.0f */, 0u)) + 1.0f * dist0;
            const bool ok = alive && !(run0 && seen0);
            seen0 = seen0 || (alive && run0);
            const uint32_t nrun6 = (run6[pi] + 64u) & (15u << 6); // run + 1, 16 -> 0
            const uint32_t lo = ((uint32_t)slot << 28) | nrun6 | (uint32_t)pi; // kind 0
            e[wire] = ((uint64_t)sel_u32(ok, f2u(cost), kNoState) << 32) | lo;
        }
        // stable sort by cost == sort by (cost bits, slot); the 35 compare-exchanges of the optimal 11-input network, three of
        // which keep only their minimum: the eight smallest are all that is looked at (tools/trellis_network_check.py)
        PIXO_CE(0, 9) PIXO_CE(1, 6) PIXO_CE(2, 4) PIXO_CE(3, 7) PIXO_CE(5, 8)
        PIXO_CE(0, 1) PIXO_CE(3, 5) PIXO_CE(4, 10) PIXO_CE(6, 9) PIXO_CE(7, 8)
        PIXO_CE(1, 3) PIXO_CE(2, 5) PIXO_CE(4, 7) PIXO_CE(8, 10)
        PIXO_CE(0, 4) PIXO_CE(1, 2) PIXO_CE(3, 7) PIXO_CE(5, 9) PIXO_CE(6, 8)
        PIXO_CE(0, 1) PIXO_CE(2, 6) PIXO_CE(4, 5) PIXO_CE(7, 8) PIXO_CE_LO(9, 10)
        PIXO_CE(2, 4) PIXO_CE(3, 6) PIXO_CE(5, 7) PIXO_CE_LO(8, 9)
        PIXO_CE(1, 2) PIXO_CE(3, 4) PIXO_CE(5, 6) PIXO_CE_LO(7, 8)
        PIXO_CE(2, 3) PIXO_CE(4, 5) PIXO_CE(6, 7)
        // back-pointers: one byte per survivor — bits 0..5 (kind, parent), bits 6..7 the low bits of the run (ignored when read)
        uint32_t lo8[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            lo8[i] = (uint32_t)e[i];
            cc[i] = (uint32_t)(e[i] >> 32);
            run6[i] = lo8[i] & (15u << 6);
        }
        env.trail_put(zz - 1, ((uint64_t)low_bytes(lo8[4], lo8[5], lo8[6], lo8[7]) << 32) | low_bytes(lo8[0], lo8[1], lo8[2], lo8[3]));
    }
    // trailing zeros: an EOB will be coded (trellis.rs:172-178); min_by: the first of equal minima
    int idx = 0;
    float best = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float c = u2f(cc[i]);
        if (run6[i] > 0) c += 4.0f;
        if (i == 0 || c < best) { best = c; idx = i; }
    }
    coef_next = env.coef(63); step_next = env.step(63);
    uint64_t trail_next = env.trail_get(62);
    for (int zz = 63; zz >= 1; zz--) {
        const uint32_t f = (uint32_t)(trail_next >> (8 * idx)) & 63u;
        const int kind = (int)(f >> 3);
        const Kinds k = candidate_kinds(coef_next / step_next);
        if (zz > 1) { coef_next = env.coef(zz - 1); step_next = env.step(zz - 1); trail_next = env.trail_get(zz - 2); }
        int v = 0;
#pragma unroll
        for (int j = 1; j < 5; j++) v = kind == j ? k.v[j] : v;
        env.out(zz, (int16_t)v);
        idx = (int)(f & 7u);
    }
}
#undef PIXO_CE
#undef PIXO_CE_LO

// ---- the same search with a block's eight survivors on EIGHT LANES (round 5; jpeg_trellis.hip trellis_lanes_kernel) ----------
// What ONE LANE (survivor s of its block) does between the group's exchanges; the kernel supplies the exchanges (the minimum over
// the eight lanes, a ballot, the zero keys through LDS, the scatter by rank), tests/emu runs the same functions lane by lane on
// the host with arrays in their place.  Keys, costs, slots and ties are quantize_block_fast's.
struct LanePre { float dist[3]; float dist0; uint32_t meta; }; // one position: the candidates' distortions, the zero candidate's;
                                                               // per candidate a byte: size << 4 | kind << 1 | valid
PIXO_TDEV LanePre lanes_prepare(float coef, float qq)
{ // (does not depend on the search: worked out for all 63 positions up front, side by side)
    const Kinds3 k = candidate_kinds3(coef / qq);
    LanePre p;
    p.meta = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float rec = (float)k.v[j] * qq, d = coef - rec;
        p.dist[j] = d * d;
        p.meta |= (((uint32_t)size_category(k.v[j]) << 4) | (k.kind[j] << 1) | (k.ok[j] ? 1u : 0u)) << (8 * j);
    }
    p.dist0 = coef * coef;
    return p;
}
// candidate j against THIS lane's parent: (cost bits, parent) — the group's minimum of these is the first strict minimum
PIXO_TDEV uint64_t lanes_cost_key(uint32_t cc, uint32_t run6, const LanePre &p, int j, int s, const float *rate_table)
{
    const uint32_t m = p.meta >> (8 * j);
    const float rate = rate_table[(run6 >> 2) | ((m >> 4) & 15u)];
    const float cost = u2f(cc) + rate + 1.0f * p.dist[j];
    return ((uint64_t)f2u(cost) << 32) | (uint32_t)s;
}
// ... and the sort key of the candidate's successor from the group's minimum
PIXO_TDEV uint64_t lanes_candidate(uint64_t bestk, const LanePre &p, int j)
{
    const uint32_t m = p.meta >> (8 * j), kind = (m >> 1) & 7u;
    const uint32_t lo = (uint32_t)bestk | (kind << 28) | (kind << 3);
    return ((uint64_t)sel_u32((m & 1u) != 0, (uint32_t)(bestk >> 32), kNoState) << 32) | lo;
}
PIXO_TDEV bool lanes_alive_run0(uint32_t cc, uint32_t run6) { return cc != kNoState && run6 == 0; }
// this lane's parent's zero successor; run0_in_front: some alive parent with run 0 sits on a lower lane of the group
PIXO_TDEV uint64_t lanes_zero_key(uint32_t cc, uint32_t run6, const LanePre &p, int s, bool run0_in_front)
{
    const bool alive = cc != kNoState, run0 = run6 == 0;
    const bool over = run6 == (15u << 6); // a ZRL symbol will be needed (trellis.rs:117-120)
    const float cost0 = u2f(cc) + u2f(sel_u32(over, 0x41200000u /* 10.0f */, 0u)) + 1.0f * p.dist0;
    const bool ok = alive && !(run0 && run0_in_front);
    const uint32_t slot = s == 0 ? 0u : 4u + (uint32_t)s;
    const uint32_t nrun6 = (run6 + 64u) & (15u << 6);
    return ((uint64_t)sel_u32(ok, f2u(cost0), kNoState) << 32) | (slot << 28) | nrun6 | (uint32_t)s;
}
// the reference's stable sort as a rank count: how many of the eleven keys lie below this lane's zero key / below `mine`
PIXO_TDEV void lanes_ranks(const uint64_t z[8], const uint64_t cand[3], uint64_t zkey, uint64_t mine, uint32_t *rank_z, uint32_t *rank_c)
{
    uint32_t rz = 0, rc = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) { rz += z[t] < zkey ? 1u : 0u; rc += z[t] < mine ? 1u : 0u; }
#pragma unroll
    for (int j = 0; j < 3; j++) { rz += cand[j] < zkey ? 1u : 0u; rc += cand[j] < mine ? 1u : 0u; }
    *rank_z = rz; *rank_c = rc;
}
// the survivor of this lane's rank: its cost, its run, its back-pointer byte ((kind, parent) in bits 0..5)
PIXO_TDEV void lanes_take(uint64_t e, uint32_t *cc, uint32_t *run6, uint8_t *back)
{
    *cc = (uint32_t)(e >> 32);
    *run6 = (uint32_t)e & (15u << 6);
    *back = (uint8_t)e;
}
// trailing zeros: an EOB will be coded (trellis.rs:172-178); the group's minimum of these keys is min_by's first minimum
PIXO_TDEV uint64_t lanes_final_key(uint32_t cc, uint32_t run6, int s)
{
    float c = u2f(cc);
    if (run6 > 0) c += 4.0f;
    return ((uint64_t)f2u(c) << 32) | (uint32_t)s;
}
// the value at one position from the kind its back-pointer names (position 0: the DC, plain rounding, trellis.rs:75)
PIXO_TDEV int lanes_value(float fq, int kind, bool dc)
{
    if (dc) return to_i16(__builtin_roundf(fq));
    const Kinds kk = candidate_kinds(fq);
    int v = 0;
#pragma unroll
    for (int j = 1; j < 5; j++) v = kind == j ? kk.v[j] : v;
    return v;
}

} // namespace pixo_trellis
