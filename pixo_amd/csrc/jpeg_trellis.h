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

} // namespace pixo_trellis
