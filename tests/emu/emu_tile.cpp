// emu_tile.cpp — TEST HARNESS.  Runs the device tile code of pixo_amd/csrc/jpeg_tile.h on
// the CPU: one workgroup = a loop over 256 "lanes" per barrier-delimited phase, LDS = a
// byte array.  The arithmetic executed is the kernel's own source (compiled with
// -DPIXO_EMU, -ffp-contract=off), so CPU-only test runs check the real kernel logic —
// lane mapping, LDS layout, swizzles, colour math, DCT order, quantiser fast path —
// against the oracle.  Not shipped; the product never links this.
#define PIXO_EMU 1
#include "../../pixo_amd/csrc/jpeg_tile.h"
#include "../../pixo_amd/csrc/jpeg_host.hpp"

#include <vector>

using namespace pixo_tile;

template <int MODE, int LOAD, bool PACKED>
static void run_image(const TileCtx &c, uint32_t tiles_x, uint32_t tiles_y, long *stats, int wave_order)
{
    typedef Geo<MODE> G;
    static_assert(lds_bytes<MODE>() <= 160 * 1024 / 6, "six workgroups must fit a CU's LDS");
    alignas(16) static uint8_t lds[lds_bytes<MODE>()];
    std::vector<float> v((size_t)192 * 64);
    uint32_t regs[G::items * G::item_regs];
    for (uint32_t ty = 0; ty < tiles_y; ty++)
        for (uint32_t tx = 0; tx < tiles_x; tx++) {
            if (stats) stats[LOAD == L_ALIGNED ? 0 : (LOAD == L_BYTES ? 1 : 2)]++; // tiles by aligned loads / byte gathers / funnel loads
            uint8_t *planar = lds;
            memset(planar, 0xA5, G::planar); // nothing may rely on a previous tile's samples
            // producer wavefront: every lane loads and converts its items of this tile
            for (int lane = 0; lane < 64; lane++) {
                LaneAddr la{};
                if (LOAD != L_BYTES) la = lane_addr<MODE>(c, tx, ty, lane);
                for (int k = 0; k < G::items; k++) producer_load_item<MODE, LOAD>(c, la, tx, ty, k, lane, &regs[k * G::item_regs]);
                for (int k = 0; k < G::items; k++) {
                    producer_fix_item<MODE, LOAD>(c, tx, k, lane, &regs[k * G::item_regs]);
                    producer_color_item<MODE, LOAD != L_BYTES>(k, lane, &regs[k * G::item_regs], planar); // (the shipped kernels: dot4 conversion with the vector loads)
                }
            }
            // (barrier) consumer wavefronts, in a caller-chosen order: they share nothing
            for (int k = 0; k < 3; k++) {
                const int w = wave_order == 0 ? k : (wave_order == 1 ? 2 - k : (k + 1) % 3);
                // a wavefront runs these steps in lockstep: every lane finishes a step before any
                // lane starts the next (the stage is written by block and read back by chunk)
                for (int l = 0; l < 64; l++) consumer_rows<MODE, PACKED>(w, l, planar, &v[(w * 64 + l) * 64]);
                for (int l = 0; l < 64; l++) consumer_cols<PACKED>(&v[(w * 64 + l) * 64]);
                // the shipped write-out: whole block into registers, two rounds of 32 blocks through the stage
                static thread_local uint32_t qw[64 * 32];
                for (int l = 0; l < 64; l++) consumer_quant<MODE, PACKED>(w, l, c.qt, &v[(w * 64 + l) * 64], &qw[l * 32]);
                for (int h = 0; h < 2; h++) {
                    for (int l = 0; l < 64; l++) consumer_stage_blocks(l, h, &qw[l * 32], lds + stage_offset<MODE>(w));
                    for (int l = 0; l < 64; l++) consumer_store_blocks<MODE>(c, tx, ty, w, l, h, lds + stage_offset<MODE>(w));
                }
            }
        }
}

extern "C" int emu_jpeg_coeffs(const uint8_t *px, uint32_t W, uint32_t H, int color_type, int subsampling,
                               int quality, int16_t *y, int16_t *cb, int16_t *cr, int allow_fast,
                               long *stats /* [tiles by aligned loads, by byte gathers, by funnel loads] or NULL */, int wave_order)
{
    float qt[pixo_host::kDeviceQtFloats];
    pixo_host::fill_device_qt((uint8_t)quality, qt);
    const bool gray = color_type == 0, s420 = !gray && subsampling == 1;
    TileCtx c;
    c.px = px; c.y = y; c.cb = cb; c.cr = cr; c.qt = qt; c.W = W; c.H = H;
    const uint32_t unit = s420 ? 16 : 8;
    c.units_x = (W + unit - 1) / unit;
    c.units_y = (H + unit - 1) / unit;
    const size_t row_bytes = (size_t)W * (gray ? 1 : 3);
    c.px_first = px; c.px_end = px + row_bytes * H;
    // the launcher's choice (jpeg_kernels.hip launch_jpeg_coeffs); allow_fast = 0 forces the byte gathers,
    // 2 forces the funnel loads even for aligned images
    const bool aligned = ((uintptr_t)px % 4 == 0) && (row_bytes % 4 == 0);
    int load = (!allow_fast || W < 4) ? L_BYTES : (aligned && allow_fast != 2 ? L_ALIGNED : L_FUNNEL);
    c.fast = load != L_BYTES;
    if (stats) stats[0] = stats[1] = stats[2] = 0;
    g_emu_px_first = c.px_first; g_emu_px_end = c.px_end; g_emu_oob_loads = 0; // (every vector load is checked against the image's bytes)
    const uint32_t tx_gray = (c.units_x + 63) / 64, ty_gray = (c.units_y + 2) / 3;
#define PIXO_RUN(MODE, TX, TY)                                                     \
    do {                                                                           \
        /* both forms of the DCT passes and the quantiser exist on the device (jpeg_tile.h block_rows): wave orders 0 and 2 run the \
           packed one, wave order 1 the scalar one — the tests use all three on every image */ \
        if (wave_order != 1) { \
            if (load == L_ALIGNED) run_image<MODE, L_ALIGNED, true>(c, TX, TY, stats, wave_order); \
            else if (load == L_FUNNEL) run_image<MODE, L_FUNNEL, true>(c, TX, TY, stats, wave_order); \
            else run_image<MODE, L_BYTES, true>(c, TX, TY, stats, wave_order); \
        } else { \
            if (load == L_ALIGNED) run_image<MODE, L_ALIGNED, false>(c, TX, TY, stats, wave_order); \
            else if (load == L_FUNNEL) run_image<MODE, L_FUNNEL, false>(c, TX, TY, stats, wave_order); \
            else run_image<MODE, L_BYTES, false>(c, TX, TY, stats, wave_order); \
        } \
    } while (0)
    if (gray) PIXO_RUN(MGRAY, tx_gray, ty_gray);
    else if (s420) PIXO_RUN(M420, (c.units_x + 31) / 32, c.units_y);
    else PIXO_RUN(M444, (c.units_x + 63) / 64, c.units_y);
#undef PIXO_RUN
    g_emu_px_first = g_emu_px_end = nullptr;
    return g_emu_oob_loads ? -(int)(g_emu_oob_loads > 1000000 ? 1000000 : g_emu_oob_loads) : 0; // < 0: loads outside the image's bytes
}

// The bracketing reciprocals of one divisor exactly as the device tables hold them (fill_device_qt
// computes them per quality; here: any q in 1..255).
static void bracket_of(int q, float *lo, float *hi)
{
    const double w_lo = (1.0 / q) * (1.0 - 0x1p-24 - 0x1p-30), w_hi = (1.0 / q) * (1.0 + 0x1p-24 + 0x1p-30);
    float l = (float)w_lo, h = (float)w_hi;
    while ((double)l > w_lo) l = nextafterf(l, 0.0f);
    while ((double)h < w_hi) h = nextafterf(h, 2.0f);
    *lo = l; *hi = h;
}

// the table block of one quality, as the device gets it (checked against bracket_of by the tests)
extern "C" void emu_device_qt(int quality, float *out /* kDeviceQtFloats */) { pixo_host::fill_device_qt((uint8_t)quality, out); }
extern "C" void emu_bracket_of(int q, float *lo, float *hi) { bracket_of(q, lo, hi); }

// quantiser fast path vs the reference formula on caller-provided values (for the
// dense/exhaustive sweeps in tests/test_quant_exact.py)
extern "C" long emu_quant_mismatches(const float *x, long n, int qlo, int qhi)
{
    long bad = 0;
    for (int q = qlo; q <= qhi; q++) {
        const float fq = (float)q;
        float lo, hi;
        bracket_of(q, &lo, &hi);
        QPair rr[8];
        float qq[8], xx[8];
        for (int i = 0; i < 8; i++) { rr[i].lo = lo; rr[i].hi = hi; qq[i] = fq; }
        for (long i = 0; i + 8 <= n; i += 8) {
            uint32_t out[4];
            for (int k = 0; k < 8; k++) xx[k] = x[i + k];
            quant_row8<false>(xx, rr, as_qtab(qq), 1.0f, out);
            for (int k = 0; k < 8; k++) {
                int16_t got = (int16_t)(out[k >> 1] >> (16 * (k & 1)));
                float want = roundf(xx[k] / fq);
                if ((float)got != want) bad++;
            }
        }
    }
    return bad;
}

// fast path ONLY (no exact fallback): returns how many inputs the safety test flags and
// how many unflagged inputs would have been wrong (must be 0).
extern "C" void emu_quant_fastpath_audit(const float *x, long n, int q, long *flagged, long *wrong_unflagged)
{
    const float fq = (float)q;
    float lo, hi;
    bracket_of(q, &lo, &hi);
    long f = 0, w = 0;
    for (long i = 0; i < n; i++) {
        float sm;
        const bool risky = quant_bracket(x[i], lo, hi, &sm) != 0; // the kernel's own test (jpeg_tile.h)
        int16_t got = (int16_t)(__builtin_bit_cast(uint32_t, sm) & 0xFFFF);
        if (risky) f++;
        else if ((float)got != roundf(x[i] / fq)) w++;
    }
    *flagged = f;
    *wrong_unflagged = w;
}

// Exhaustive audit over every f32 bit pattern in [lo_bits, hi_bits] (both signs), one q.
extern "C" void emu_quant_exhaustive(int q, uint32_t lo_bits, uint32_t hi_bits, long *flagged,
                                     long *wrong_unflagged, long *wrong_final)
{
    const float fq = (float)q;
    float lo, hi;
    bracket_of(q, &lo, &hi);
    long f = 0, w = 0, wf = 0;
    for (int sign = 0; sign < 2; sign++)
        for (uint64_t b = lo_bits; b <= hi_bits; b++) {
            uint32_t u = (uint32_t)b | (sign ? 0x80000000u : 0u);
            float x;
            memcpy(&x, &u, 4);
            float want = roundf(x / fq);
            float sm;
            const bool risky = quant_bracket(x, lo, hi, &sm) != 0; // the kernel's own test (jpeg_tile.h)
            float got = (float)(int16_t)(__builtin_bit_cast(uint32_t, sm) & 0xFFFF);
            if (risky) { f++; got = want; }
            else if (got != want) w++;
            if (got != want) wf++;
        }
    *flagged = f; *wrong_unflagged = w; *wrong_final = wf;
}

// ---------------------------------------------------------------------------------------------
// Device entropy stage (pixo_amd/csrc/jpeg_scan_block.h) lane by lane: lengths, prefix sum,
// pack into a zeroed MSB-first word stream, 1-padding, byte stuffing — the steps of
// jpeg_entropy.hip with the workgroup collectives replaced by loops.
// ---------------------------------------------------------------------------------------------
#include "../../pixo_amd/csrc/jpeg_scan_block.h"

extern "C" void emu_scan_tables(const int16_t *y, const int16_t *cb, const int16_t *cr, uint32_t w, uint32_t h,
                                int color_type, int subsampling, int optimize, uint32_t *out /* 536 words */)
{
    pixo_jpeg_options o{};
    o.width = w; o.height = h; o.color_type = (uint8_t)color_type; o.subsampling = (uint8_t)subsampling; o.quality = 80;
    pixo_host::HuffSet hs = pixo_host::HuffSet::standard();
    if (optimize) {
        // statistics through the DEVICE visitor (CountVisitor), tables through the host builder
        const pixo_host::Geometry g = pixo_host::geometry(w, h, o.color_type, o.subsampling);
        const int mode = g.gray ? 0 : (g.s420 ? 2 : 1);
        const uint64_t n = g.y_blocks + 2 * g.c_blocks;
        std::vector<uint32_t> hist(pixo_scan::kTableWords, 0);
        for (uint64_t s = 0; s < n; s++) {
            const pixo_scan::BlockRef r = pixo_scan::block_of(mode, s);
            const int16_t *base = r.comp == 0 ? y : (r.comp == 1 ? cb : cr);
            uint32_t wds[32];
            memcpy(wds, base + r.index * 64, 128);
            pixo_scan::CountVisitor v{hist.data() + (r.comp ? 1 : 0) * pixo_scan::kClassSyms};
            pixo_scan::walk_block(wds, r.index ? base[(r.index - 1) * 64] : 0, v);
        }
        uint64_t dc[2][12], ac[2][256];
        for (int c = 0; c < 2; c++) {
            for (int i = 0; i < 12; i++) dc[c][i] = hist[c * 268 + i];
            for (int i = 0; i < 256; i++) ac[c][i] = hist[c * 268 + 12 + i];
        }
        hs = pixo_host::HuffSet::optimized(dc, ac, !g.gray);
    }
    pixo_host::pack_scan_tables(hs, out);
}

// Symbol statistics: the reference-shaped walk with CountVisitor against the flat walk of scan_count_kernel
// (block_count_flat into walk-table slots, mapped back with walk_slot_symbol).  Returns the number of counters that
// differ (0 = same histograms); out (536 words, may be null) receives the flat walk's histogram.
extern "C" long emu_count_compare(const int16_t *y, const int16_t *cb, const int16_t *cr, int mode, uint64_t nblocks, uint32_t *out)
{
    using namespace pixo_scan;
    std::vector<uint32_t> want(kTableWords, 0), slots(kWalkWords + 1, 0), got(kTableWords, 0);
    struct EmuBump {
        uint32_t *hist;
        void bump(uint32_t slot, bool on, uint32_t amount) { if (on) hist[slot] += amount; }
    };
    for (uint64_t s = 0; s < nblocks; s++) {
        const BlockRef r = block_of(mode, s);
        const int16_t *base = r.comp == 0 ? y : (r.comp == 1 ? cb : cr);
        uint32_t wds[32];
        memcpy(wds, base + r.index * 64, 128);
        const int prev = r.index ? base[(r.index - 1) * 64] : 0, cls = r.comp ? 1 : 0;
        CountVisitor v{want.data() + cls * kClassSyms};
        walk_block(wds, prev, v);
        EmuBump h{slots.data() + cls * kWalkClassWords};
        block_count_flat(wds, prev, h);
    }
    long lost = 0;
    for (int i = 0; i < kWalkWords; i++) {
        const int sym = walk_slot_symbol(i % kWalkClassWords);
        if (sym >= 0) got[(i / kWalkClassWords) * kClassSyms + sym] += slots[i];
        else lost += slots[i]; // a count in a slot that stands for no symbol
    }
    long bad = lost;
    for (int i = 0; i < kTableWords; i++) bad += want[i] != got[i];
    if (out) memcpy(out, got.data(), kTableWords * 4);
    return bad;
}

// The branch-free walkers of the single-pass kernels (block_pack_flat, jpeg_scan_fused.hip) in
// the same harness: lengths, prefix sum, packing in reverse block order into a zeroed stream, padding, stuffing.
struct EmuOrSink {
    uint32_t *stream;
    uint64_t first_word; // the packer's word index is relative to this
    void or_word(bool flush, uint32_t word, uint32_t value) { if (flush) stream[first_word + word] |= value; }
};
extern "C" long emu_scan_flat(const int16_t *y, const int16_t *cb, const int16_t *cr, int mode, uint64_t nblocks,
                              const uint32_t *tables, uint8_t *out, long cap)
{
    using namespace pixo_scan;
    std::vector<uint32_t> len(nblocks);
    std::vector<uint64_t> off(nblocks);
    uint32_t wtab[kWalkWords]; // the flat walk's own table form, built as the kernel builds it in LDS
    for (int i = 0; i < kWalkWords; i++) wtab[i] = walk_table_word(tables, i);
    struct NullSink { void or_word(bool, uint32_t, uint32_t) {} };
    auto load = [&](uint64_t s, uint32_t *wds, int *prev, int *cls) {
        const BlockRef r = block_of(mode, s);
        const int16_t *base = r.comp == 0 ? y : (r.comp == 1 ? cb : cr);
        memcpy(wds, base + r.index * 64, 128);
        *prev = r.index ? base[(r.index - 1) * 64] : 0;
        *cls = r.comp ? 1 : 0;
    };
    uint64_t total = 0;
    for (uint64_t s = 0; s < nblocks; s++) {
        uint32_t wds[32]; int prev, cls;
        load(s, wds, &prev, &cls);
        FlatPack<NullSink> q;
        q.acc = 0; q.pending = 0; q.word = 0;
        block_pack_flat(wds, prev, wtab + cls * kWalkClassWords, q);
        len[s] = q.word * 32u + q.pending;
        off[s] = total; total += len[s];
    }
    std::vector<uint32_t> stream(total / 32 + 2, 0);
    for (uint64_t i = nblocks; i-- > 0;) {
        uint32_t wds[32]; int prev, cls;
        load(i, wds, &prev, &cls);
        FlatPack<EmuOrSink> p;
        p.sink = EmuOrSink{stream.data(), off[i] >> 5};
        p.acc = 0; p.pending = (uint32_t)(off[i] & 31); p.word = 0;
        block_pack_flat(wds, prev, wtab + cls * kWalkClassWords, p);
        p.finish();
        if (p.word * 32ull + p.pending != (off[i] & 31) + len[i]) return -2; // the walk into nothing and the walk into the stream must agree on the length
    }
    const int n = (int)((8 - (total & 7)) & 7);
    if (n) stream[total >> 5] |= ((1u << n) - 1u) << (32 - (int)(total & 31) - n);
    const uint64_t nbytes = (total + 7) / 8;
    long o = 0;
    for (uint64_t b = 0; b < nbytes; b++) {
        const uint8_t byte = (uint8_t)(stream[b >> 2] >> (24 - 8 * (b & 3)));
        if (o + 2 > cap) return -1;
        out[o++] = byte;
        if (byte == 0xFF) out[o++] = 0x00;
    }
    return o;
}

// A scan coded in PIECES (pixo_dev::ScanPiece, pieces.cpp device_entropy_pieces) on the CPU: piece k codes blocks
// [k * per_piece, ...) with the flat walk into a stream of its own that starts with the `lead` = (bits before the piece) % 8
// last bits of the piece before — taken from that piece's stream, like the device does — so that it is byte-aligned with
// the scan; every piece but the last is stuffed in whole bytes only, the last one is padded with 1-bits; the stuffed
// pieces are concatenated.  The result must be the oracle's scan.
extern "C" long emu_scan_pieces(const int16_t *y, const int16_t *cb, const int16_t *cr, int mode, uint64_t nblocks,
                                const uint32_t *tables, uint64_t per_piece, uint8_t *out, long cap)
{
    using namespace pixo_scan;
    uint32_t wtab[kWalkWords];
    for (int i = 0; i < kWalkWords; i++) wtab[i] = walk_table_word(tables, i);
    std::vector<uint32_t> prev_stream;
    uint64_t before_prev = 0, before = 0; // chain[k - 1], chain[k]
    long o = 0;
    const uint64_t pieces = (nblocks + per_piece - 1) / per_piece;
    for (uint64_t k = 0; k < pieces; k++) {
        const uint64_t first = k * per_piece, n = std::min<uint64_t>(per_piece, nblocks - first);
        const uint32_t lead = (uint32_t)(before & 7);
        std::vector<uint32_t> stream(n * 53 + 4, 0);
        if (k && lead) { // the last, partial byte of the previous piece's stream
            const uint64_t prev_bits = (before_prev & 7) + (before - before_prev), at = prev_bits >> 3;
            const uint32_t byte = (prev_stream[at >> 2] >> (24 - 8 * (uint32_t)(at & 3))) & 0xFFu;
            stream[0] = (byte & (0xFF00u >> lead) & 0xFFu) << 24;
        }
        uint64_t pos = lead;
        for (uint64_t s = first; s < first + n; s++) {
            const BlockRef r = block_of(mode, s);
            const int16_t *base = r.comp == 0 ? y : (r.comp == 1 ? cb : cr);
            uint32_t wds[32];
            memcpy(wds, base + r.index * 64, 128);
            FlatPack<EmuOrSink> p;
            p.sink = EmuOrSink{stream.data(), pos >> 5};
            p.acc = 0; p.pending = (uint32_t)(pos & 31); p.word = 0;
            block_pack_flat(wds, r.index ? base[(r.index - 1) * 64] : 0, wtab + (r.comp ? 1 : 0) * kWalkClassWords, p);
            p.finish();
            pos = ((pos >> 5) + p.word) * 32 + p.pending;
        }
        const bool last = k + 1 == pieces;
        uint64_t total = pos; // the stream's length with its leading bits
        if (last) {
            const int pad = (int)((8 - (total & 7)) & 7);
            if (pad) stream[total >> 5] |= ((1u << pad) - 1u) << (32 - (int)(total & 31) - pad);
            total += pad;
        }
        const uint64_t nbytes = total / 8; // (whole bytes: every piece but the last leaves its last bits to the next)
        for (uint64_t b = 0; b < nbytes; b++) {
            const uint8_t byte = (uint8_t)(stream[b >> 2] >> (24 - 8 * (b & 3)));
            if (o + 2 > cap) return -1;
            out[o++] = byte;
            if (byte == 0xFF) out[o++] = 0x00;
        }
        before_prev = before;
        before += pos - lead;
        prev_stream.swap(stream);
    }
    return o;
}

extern "C" long emu_scan(const int16_t *y, const int16_t *cb, const int16_t *cr, int mode, uint64_t nblocks,
                         const uint32_t *tables, uint8_t *out, long cap)
{
    using namespace pixo_scan;
    std::vector<uint32_t> len(nblocks);
    std::vector<uint64_t> off(nblocks);
    auto load = [&](uint64_t s, uint32_t *wds, int *prev, int *cls) {
        const BlockRef r = block_of(mode, s);
        const int16_t *base = r.comp == 0 ? y : (r.comp == 1 ? cb : cr);
        memcpy(wds, base + r.index * 64, 128);
        *prev = r.index ? base[(r.index - 1) * 64] : 0;
        *cls = r.comp ? 1 : 0;
    };
    uint64_t total = 0;
    for (uint64_t s = 0; s < nblocks; s++) {
        uint32_t wds[32]; int prev, cls;
        load(s, wds, &prev, &cls);
        LengthVisitor v{tables + cls * kClassSyms, 0};
        walk_block(wds, prev, v);
        len[s] = v.bits; off[s] = total; total += v.bits;
    }
    std::vector<uint32_t> stream(total / 32 + 2, 0);
    // blocks packed in REVERSE order: the result must not depend on which lane runs first
    for (uint64_t i = nblocks; i-- > 0;) {
        uint32_t wds[32]; int prev, cls;
        load(i, wds, &prev, &cls);
        PackVisitor v;
        v.tab = tables + cls * kClassSyms;
        v.begin(stream.data(), off[i]);
        walk_block(wds, prev, v);
        v.finish();
        if (i == nblocks - 1) {
            const int n = (int)((8 - (total & 7)) & 7);
            if (n) v.or_word(total >> 5, ((1u << n) - 1u) << (32 - (int)(total & 31) - n));
        }
    }
    const uint64_t nbytes = (total + 7) / 8;
    long o = 0;
    for (uint64_t b = 0; b < nbytes; b++) {
        const uint8_t byte = (uint8_t)(stream[b >> 2] >> (24 - 8 * (b & 3)));
        if (o + 2 > cap) return -1;
        out[o++] = byte;
        if (byte == 0xFF) out[o++] = 0x00;
    }
    return o;
}


// ---------------------------------------------------------------------------------------------
// Device trellis quantiser (pixo_amd/csrc/jpeg_trellis.h) on the host: n blocks of 64 floats.
// ---------------------------------------------------------------------------------------------
#include "../../pixo_amd/csrc/jpeg_trellis.h"
extern "C" void emu_trellis(const float *dct, const float *q, long nblocks, int16_t *out)
{
    for (long b = 0; b < nblocks; b++) {
        uint32_t trail[63 * 8];
        uint8_t counts[63];
        pixo_trellis::quantize_block(dct + b * 64, q, out + b * 64, trail, counts);
    }
}

// The register-resident restatement the kernel actually runs (quantize_block_fast), same interface.
namespace {
struct HostTrellisEnv {
    const float *dct, *q;
    int16_t *res;
    uint64_t trail[63];
    float table[256];
    float coef(int zz) const { return dct[pixo_trellis::kZigzagNat[zz]]; }
    float step(int zz) const { return q[pixo_trellis::kZigzagNat[zz]]; }
    float rate_at(uint32_t byte_off) const { return table[byte_off / 4]; }
    void trail_put(int pos, uint64_t w) { trail[pos] = w; }
    uint64_t trail_get(int pos) const { return trail[pos]; }
    void out(int zz, int16_t v) { res[pixo_trellis::kZigzagNat[zz]] = v; }
};
} // namespace
extern "C" void emu_trellis_fast(const float *dct, const float *q, long nblocks, int16_t *out)
{
    HostTrellisEnv env;
    for (int rs = 0; rs < 256; rs++) env.table[rs] = pixo_trellis::rate_value(rs);
    for (long b = 0; b < nblocks; b++) {
        env.dct = dct + b * 64; env.q = q; env.res = out + b * 64;
        pixo_trellis::quantize_block_fast(env);
    }
}

// The eight-lane form of the search (trellis_lanes_kernel, jpeg_trellis.hip): the per-lane functions of jpeg_trellis.h run lane by
// lane, the group's exchanges — the minimum over the lanes, the ballot, the zero keys through LDS, the scatter by rank — as loops
// over arrays.
extern "C" void emu_trellis_lanes(const float *dct, const float *q, long nblocks, int16_t *out)
{
    using namespace pixo_trellis;
    float table[256];
    for (int rs = 0; rs < 256; rs++) table[rs] = rate_value(rs);
    for (long b = 0; b < nblocks; b++) {
        const float *d = dct + b * 64;
        int16_t *res = out + b * 64;
        LanePre pre[64];
        for (int zz = 0; zz < 64; zz++) pre[zz] = lanes_prepare(d[kZigzagNat[zz]], q[kZigzagNat[zz]]);
        uint32_t cc[8], run6[8];
        uint8_t trail[63][8];
        for (int s = 0; s < 8; s++) { cc[s] = s == 0 ? 0u : kNoState; run6[s] = 0; }
        for (int zz = 1; zz < 64; zz++) {
            const LanePre &p = pre[zz];
            uint64_t cand[3];
            for (int j = 0; j < 3; j++) {
                uint64_t best = ~0ull;
                for (int s = 0; s < 8; s++) best = std::min(best, lanes_cost_key(cc[s], run6[s], p, j, s, table));
                cand[j] = lanes_candidate(best, p, j);
            }
            uint64_t z[8];
            for (int s = 0; s < 8; s++) {
                bool in_front = false;
                for (int t = 0; t < s; t++) in_front |= lanes_alive_run0(cc[t], run6[t]);
                z[s] = lanes_zero_key(cc[s], run6[s], p, s, in_front);
            }
            uint64_t sorted[12];
            for (int i = 0; i < 12; i++) sorted[i] = 0;
            for (int s = 0; s < 8; s++) {
                const uint64_t mine = s == 0 ? cand[0] : (s == 1 ? cand[1] : cand[2]);
                uint32_t rz, rc;
                lanes_ranks(z, cand, z[s], mine, &rz, &rc);
                sorted[rz] = z[s];
                sorted[s < 3 ? rc : 11u] = mine;
            }
            for (int s = 0; s < 8; s++) lanes_take(sorted[s], &cc[s], &run6[s], &trail[zz - 1][s]);
        }
        uint64_t best = ~0ull;
        for (int s = 0; s < 8; s++) best = std::min(best, lanes_final_key(cc[s], run6[s], s));
        int idx = (int)(uint32_t)best;
        int kind[64];
        kind[0] = 0;
        for (int zz = 63; zz >= 1; zz--) {
            const uint32_t f = trail[zz - 1][idx] & 63u;
            kind[zz] = (int)(f >> 3);
            idx = (int)(f & 7u);
        }
        for (int zz = 0; zz < 64; zz++) {
            const int nat = kZigzagNat[zz];
            res[nat] = (int16_t)lanes_value(d[nat] / q[nat], kind[zz], zz == 0);
        }
    }
}

// (a lane's scratch: the sink of the flat walks in the single-pass kernels)
struct EmuLaneSink {
    uint32_t *words;
    uint32_t cap;
    void or_word(bool, uint32_t word, uint32_t value) { words[word < cap ? word : cap] = value; } // (the last store wins)
};
// ---------------------------------------------------------------------------------------------
// Device progressive scan coder (jpeg_scan_block.h: band_flags, band_run_before, prog_emit) lane by
// lane: flags, rank / by_rank, lengths, prefix sum, pack (blocks in reverse order), pad, stuffing.
// out receives the seven entropy-coded segments back to back; seg_len[i] their byte counts.
// ---------------------------------------------------------------------------------------------
extern "C" long emu_progressive(const int16_t *y, const int16_t *cb, const int16_t *cr, uint64_t yb, uint64_t cbn,
                                const uint32_t *tables, uint8_t *out, long cap, long *seg_len)
{
    using namespace pixo_scan;
    ProgLayout lay;
    const uint64_t sizes[7] = {yb, cbn, cbn, yb, yb, cbn, cbn};
    lay.first[0] = 0;
    for (int i = 0; i < 7; i++) lay.first[i + 1] = lay.first[i] + sizes[i];
    const uint64_t V = lay.first[7];
    std::vector<uint32_t> flags(V, 0), by_rank(V + 1, 0), len(V);
    std::vector<uint64_t> rank(V + 1, 0), off(V + 1, 0);
    auto load = [&](uint64_t v, int *scan, uint32_t *wds, int *prev) {
        *scan = prog_scan_of(lay, v);
        const int comp = prog_comp(*scan);
        const int16_t *base = comp == 0 ? y : (comp == 1 ? cb : cr);
        const uint64_t b = v - lay.first[*scan];
        memcpy(wds, base + b * 64, 128);
        *prev = b ? base[(b - 1) * 64] : 0;
    };
    for (uint64_t v = 0; v < V; v++) {
        int scan, prev; uint32_t wds[32];
        load(v, &scan, wds, &prev);
        if (scan >= 3) flags[v] = prog_band(scan) == 0 ? band_flags<1, 10>(wds) : (prog_band(scan) == 1 ? band_flags<11, 63>(wds) : band_flags<1, 63>(wds));
    }
    for (uint64_t v = 0; v < V; v++) { rank[v + 1] = rank[v] + (flags[v] & 1u); if (flags[v] & 1u) by_rank[rank[v]] = (uint32_t)v; }
    auto emit = [&](uint64_t v, auto &vis) {
        int scan, prev; uint32_t wds[32];
        load(v, &scan, wds, &prev);
        vis.tab = tables + (prog_comp(scan) ? 1 : 0) * kClassSyms;
        const uint32_t before = scan >= 3 ? band_run_before(v, lay.first[scan], rank[v], rank[lay.first[scan]], by_rank.data(), flags.data()) : 0;
        prog_emit(scan, wds, prev, flags[v], before, v + 1 == lay.first[scan + 1], vis);
        return scan;
    };
    for (uint64_t v = 0; v < V; v++) { LengthVisitor lv{nullptr, 0}; emit(v, lv); len[v] = lv.bits; off[v + 1] = off[v] + lv.bits; }
    long o = 0;
    for (int i = 0; i < 7; i++) {
        const uint64_t bits = off[lay.first[i + 1]] - off[lay.first[i]];
        std::vector<uint32_t> stream(bits / 32 + 2, 0);
        for (uint64_t v = lay.first[i + 1]; v-- > lay.first[i];) {
            PackVisitor pv;
            pv.begin(stream.data(), off[v] - off[lay.first[i]]);
            emit(v, pv);
            pv.finish();
        }
        const int n = (int)((8 - (bits & 7)) & 7);
        if (n) stream[bits >> 5] |= ((1u << n) - 1u) << (32 - (int)(bits & 31) - n);
        const uint64_t nbytes = (bits + 7) / 8;
        const long start = o;
        for (uint64_t b = 0; b < nbytes; b++) {
            const uint8_t byte = (uint8_t)(stream[b >> 2] >> (24 - 8 * (b & 3)));
            if (o + 2 > cap) return -1;
            out[o++] = byte;
            if (byte == 0xFF) out[o++] = 0x00;
        }
        seg_len[i] = o - start;
    }
    return o;
}

// The SINGLE-PASS form of the progressive coder (prog_code_kernel, jpeg_scan_fused.hip) group by group on the CPU: every lane
// codes its block's own symbols with the flat band walk into a scratch from bit 0 (band_pack_flat / dc_pack_flat), the run
// counter's contributions come from the ballots of each 64-lane wavefront (band_count_in_wave, band_wave_summary), the
// count carried from wavefront to wavefront and from group to group, and band_edge says what goes in front of and behind
// the lane's symbols.  Same output format as emu_progressive.
extern "C" long emu_progressive_flat(const int16_t *y, const int16_t *cb, const int16_t *cr, uint64_t yb, uint64_t cbn,
                                     const uint32_t *tables, uint8_t *out, long cap, long *seg_len)
{
    using namespace pixo_scan;
    const uint64_t sizes[7] = {yb, cbn, cbn, yb, yb, cbn, cbn};
    uint32_t wtab[kWalkWords];
    for (int i = 0; i < kWalkWords; i++) wtab[i] = walk_table_word(tables, i);
    const int kLanes = 192, kScratch = 64;
    long o = 0;
    for (int scan = 0; scan < 7; scan++) {
        const int comp = prog_comp(scan), cls = comp ? 1 : 0;
        const int16_t *base = comp == 0 ? y : (comp == 1 ? cb : cr);
        const int ss = scan < 3 ? 0 : (scan == 4 ? 11 : 1), se = scan < 3 ? 0 : (scan == 3 ? 10 : 63);
        uint32_t eob_syms[15];
        for (int n = 0; n < 15; n++) eob_syms[n] = tables[cls * kClassSyms + kDcSyms + (n << 4)];
        std::vector<uint32_t> stream(1, 0);
        uint64_t bits = 0;
        auto put = [&](uint32_t left, uint32_t len) { // `len` bits at the top of `left`
            for (uint32_t i = 0; i < len; i++) {
                if ((bits >> 5) >= stream.size()) stream.push_back(0);
                if ((left >> (31 - i)) & 1u) stream[bits >> 5] |= 1u << (31 - (bits & 31));
                bits++;
            }
        };
        uint64_t carry_group = 0; // the count flowing into the group
        for (uint64_t g0 = 0; g0 < sizes[scan]; g0 += kLanes) {
            uint32_t scratch[kLanes][kScratch + 1];
            uint32_t len[kLanes];
            bool z[kLanes], t[kLanes], live[kLanes];
            for (int l = 0; l < kLanes; l++) {
                const uint64_t b = g0 + l;
                live[l] = b < sizes[scan];
                z[l] = t[l] = false; len[l] = 0;
                if (!live[l]) continue;
                uint32_t wds[32];
                memcpy(wds, base + b * 64, 128);
                FlatPack<EmuLaneSink> p;
                p.sink = EmuLaneSink{scratch[l], (uint32_t)kScratch};
                p.acc = 0; p.pending = 0; p.word = 0;
                if (scan < 3) dc_pack_flat(wds, b ? base[(b - 1) * 64] : 0, wtab + cls * kWalkClassWords, p);
                else band_pack_flat(wds, ss, se, wtab + cls * kWalkClassWords, p, &z[l], &t[l]);
                len[l] = p.word * 32u + p.pending;
                p.finish();
            }
            uint64_t wave_in = carry_group;
            bool group_any = false;
            for (int wv = 0; wv < kLanes / 64; wv++) {
                uint64_t zmask = 0, tmask = 0;
                for (int l = 0; l < 64; l++) {
                    if (z[wv * 64 + l]) zmask |= 1ull << l;
                    if (t[wv * 64 + l] && live[wv * 64 + l]) tmask |= 1ull << l;
                }
                for (int l = 0; l < 64; l++) {
                    const int L = wv * 64 + l;
                    if (!live[L]) continue;
                    BandEdge e{};
                    if (scan >= 3) {
                        const BandCount c = band_count_in_wave(zmask, tmask, l);
                        e = band_edge((uint32_t)(c.local + (c.carried ? wave_in : 0)), true, z[L], t[L], g0 + L + 1 == sizes[scan], eob_syms);
                    }
                    put(e.pre.left, e.pre.len);
                    for (uint32_t i = 0; i < len[L]; i += 32) put(scratch[L][i >> 5], std::min(32u, len[L] - i));
                    put(e.post.left, e.post.len);
                }
                bool any; uint32_t tail;
                band_wave_summary(zmask, tmask, &any, &tail);
                wave_in = any ? tail : wave_in + tail;
                group_any |= any;
            }
            carry_group = wave_in;
            (void)group_any;
        }
        const int n = (int)((8 - (bits & 7)) & 7);
        put(0xFFFFFFFFu, (uint32_t)n);
        const uint64_t nbytes = bits / 8;
        const long start = o;
        for (uint64_t b = 0; b < nbytes; b++) {
            const uint8_t byte = (uint8_t)(stream[b >> 2] >> (24 - 8 * (b & 3)));
            if (o + 2 > cap) return -1;
            out[o++] = byte;
            if (byte == 0xFF) out[o++] = 0x00;
        }
        seg_len[scan] = o - start;
    }
    return o;
}

// Round 5's form of the same kernel: a group holds 192 blocks of ONE component and walks each block ONCE — the DC symbol, and
// all AC bands of the component in one walk into one scratch (bands_pack_flat: luminance 1..10 and 11..63 back to back,
// chrominance 1..63); every scan's bits are then gathered from its part of the scratch (scratch_bits_word) between what the
// scan's own run counter adds.  Same output format as emu_progressive.
extern "C" long emu_progressive_one_walk(const int16_t *y, const int16_t *cb, const int16_t *cr, uint64_t yb, uint64_t cbn,
                                         const uint32_t *tables, uint8_t *out, long cap, long *seg_len)
{
    using namespace pixo_scan;
    uint32_t wtab[kWalkWords];
    for (int i = 0; i < kWalkWords; i++) wtab[i] = walk_table_word(tables, i);
    const int kLanes = 192, kScratch = 64;
    struct Stream {
        std::vector<uint32_t> words{0};
        uint64_t bits = 0;
        void put(uint32_t left, uint32_t len)
        {
            for (uint32_t i = 0; i < len; i++) {
                if ((bits >> 5) >= words.size()) words.push_back(0);
                if ((left >> (31 - i)) & 1u) words[bits >> 5] |= 1u << (31 - (bits & 31));
                bits++;
            }
        }
    } streams[7];
    for (int comp = 0; comp < 3; comp++) {
        const int cls = comp ? 1 : 0;
        const int16_t *base = comp == 0 ? y : (comp == 1 ? cb : cr);
        const uint64_t nblocks = comp == 0 ? yb : cbn;
        const bool split = comp == 0;
        const int dc_scan = comp, band_scan[2] = {comp == 0 ? 3 : 4 + comp, comp == 0 ? 4 : -1};
        uint32_t eob_syms[15];
        for (int n = 0; n < 15; n++) eob_syms[n] = tables[cls * kClassSyms + kDcSyms + (n << 4)];
        uint64_t carry_group[2] = {0, 0};
        for (uint64_t g0 = 0; g0 < nblocks; g0 += kLanes) {
            uint32_t scratch[kLanes][kScratch + 1], dcw[kLanes][2];
            uint32_t dclen[kLanes], first[kLanes], total[kLanes];
            bool z[2][kLanes], t[2][kLanes], live[kLanes];
            for (int l = 0; l < kLanes; l++) {
                const uint64_t b = g0 + l;
                live[l] = b < nblocks;
                z[0][l] = z[1][l] = t[0][l] = t[1][l] = false; dclen[l] = first[l] = total[l] = 0;
                if (!live[l]) continue;
                uint32_t wds[32];
                memcpy(wds, base + b * 64, 128);
                {
                    FlatPack<EmuLaneSink> p;
                    p.sink = EmuLaneSink{dcw[l], 1u};
                    p.acc = 0; p.pending = 0; p.word = 0;
                    dc_pack_flat(wds, b ? base[(b - 1) * 64] : 0, wtab + cls * kWalkClassWords, p);
                    dclen[l] = p.word * 32u + p.pending;
                    p.finish();
                }
                FlatPack<EmuLaneSink> p;
                p.sink = EmuLaneSink{scratch[l], (uint32_t)kScratch};
                p.acc = 0; p.pending = 0; p.word = 0;
                bool any[2], ez[2];
                bands_pack_flat(wds, split, wtab + cls * kWalkClassWords, p, &first[l], any, ez);
                total[l] = p.word * 32u + p.pending;
                p.finish();
                for (int s = 0; s < 2; s++) { z[s][l] = any[s]; t[s][l] = ez[s]; }
            }
            for (int l = 0; l < kLanes; l++)
                if (live[l]) streams[dc_scan].put(dcw[l][0], dclen[l]);
            for (int s = 0; s < 2; s++) {
                if (band_scan[s] < 0) continue;
                Stream &st = streams[band_scan[s]];
                uint64_t wave_in = carry_group[s];
                for (int wv = 0; wv < kLanes / 64; wv++) {
                    uint64_t zmask = 0, tmask = 0;
                    for (int l = 0; l < 64; l++) {
                        if (z[s][wv * 64 + l]) zmask |= 1ull << l;
                        if (t[s][wv * 64 + l] && live[wv * 64 + l]) tmask |= 1ull << l;
                    }
                    for (int l = 0; l < 64; l++) {
                        const int L = wv * 64 + l;
                        if (!live[L]) continue;
                        const BandCount c = band_count_in_wave(zmask, tmask, l);
                        const BandEdge e = band_edge((uint32_t)(c.local + (c.carried ? wave_in : 0)), true, z[s][L], t[s][L], g0 + L + 1 == nblocks, eob_syms);
                        st.put(e.pre.left, e.pre.len);
                        const uint32_t from = s ? first[L] : 0u, n = s ? total[L] - first[L] : first[L];
                        for (uint32_t j = 0; 32u * j < n; j++) st.put(scratch_bits_word(scratch[L], (uint32_t)kScratch, from, n, j), std::min(32u, n - 32u * j));
                        st.put(e.post.left, e.post.len);
                    }
                    bool any; uint32_t tail;
                    band_wave_summary(zmask, tmask, &any, &tail);
                    wave_in = any ? tail : wave_in + tail;
                }
                carry_group[s] = wave_in;
            }
        }
    }
    long o = 0;
    for (int scan = 0; scan < 7; scan++) {
        Stream &st = streams[scan];
        const uint64_t blocks = prog_comp(scan) == 0 ? yb : cbn;
        const long start = o;
        if (blocks) {
            const int n = (int)((8 - (st.bits & 7)) & 7);
            st.put(0xFFFFFFFFu, (uint32_t)n);
            const uint64_t nbytes = st.bits / 8;
            for (uint64_t b = 0; b < nbytes; b++) {
                const uint8_t byte = (uint8_t)(st.words[b >> 2] >> (24 - 8 * (b & 3)));
                if (o + 2 > cap) return -1;
                out[o++] = byte;
                if (byte == 0xFF) out[o++] = 0x00;
            }
        }
        seg_len[scan] = o - start;
    }
    return o;
}

// tables for the progressive coder: like pack_scan_tables, absent symbols = the (0, 4 bits) fallback
extern "C" void emu_progressive_tables(uint32_t *out /* 536 words */)
{
    pixo_host::pack_scan_tables(pixo_host::HuffSet::standard(), out);
    for (int i = 0; i < pixo_scan::kTableWords; i++) if ((out[i] >> 16) == 0) out[i] = 4u << 16;
}

// ---------------------------------------------------------------------------------------------
// PNG row filters: the kernel's per-group arithmetic (pixo_amd/csrc/png_filter_math.h) driven row by row
// and group by group on the host — neighbour alignment for every pixel size, SWAR filters, packed Paeth,
// scores, the reference's decision sequences, checksum terms and their combination.  `strategy` is the
// one the launcher would run (the <= 4096-pixel rule and the sequential AdaptiveFast are host logic in
// png_api.cpp).  Returns the Adler-32 of the stream.
// ---------------------------------------------------------------------------------------------
#include "../../pixo_amd/csrc/png_filter_math.h"
namespace {
uint32_t host_dword(const uint8_t *row, long k, long n)
{ // bytes [4k, 4k + 4) of the row, zero outside it (what load_dword / load_six deliver)
    uint32_t v = 0;
    if (!row || k < 0) return 0;
    for (int b = 0; b < 4; b++)
        if (4 * k + b < n) v |= (uint32_t)row[4 * k + b] << (8 * b);
    return v;
}
template <int BPP> uint32_t emu_png_rows(const uint8_t *data, long n, long height, int strategy, uint8_t *out)
{
    using namespace pixo_png;
    const uint64_t M = 65521;
    uint64_t a1 = 1, a2 = 0; // Adler state over the whole stream
    const long ndw = (n + 3) / 4;
    for (long y = 0; y < height; y++) {
        const uint8_t *row = data + y * n, *prev = y ? row - n : nullptr;
        int f = strategy;
        if (strategy == S_BIGRAMS) { // five passes over the row with a "pair seen" bitmap, like bigram_score
            unsigned long long tot[5];
            static thread_local uint32_t bitmap[2048];
            for (int cand = F_NONE; cand <= F_PAETH; cand++) {
                memset(bitmap, 0, sizeof bitmap);
                auto dword = [&](long k0, int j) {
                    Raw r;
                    for (int i = 0; i < 6; i++) { r.x[i] = host_dword(row, k0 - 2 + i, n); r.u[i] = host_dword(prev, k0 - 2 + i, n); }
                    Group g;
                    group_of<BPP, true>(r, (int)k0, (int)n, g);
                    return filtered(cand, g, j);
                };
                for (long k0 = 0; k0 < ndw; k0 += 4) {
                    uint32_t v[5];
                    for (int j = 0; j < 4; j++) v[j] = dword(k0, j);
                    v[4] = k0 + 4 < ndw ? dword(k0 + 4, 0) : 0;
                    for (int j = 0; j < 4; j++) {
                        uint32_t key[4];
                        const int cnt = bigram_keys(v[j], v[j + 1], (int)(n - 1 - 4 * (k0 + j)), key);
                        for (int i = 0; i < cnt; i++) bitmap[key[i] >> 5] |= 1u << (key[i] & 31u);
                    }
                }
                unsigned long long count = 0;
                for (int i = 0; i < 2048; i++) count += (unsigned)__builtin_popcount(bitmap[i]);
                tot[cand] = count;
            }
            f = decide_bigrams(tot);
        } else if (strategy > S_PAETH) {
            uint32_t sc[5] = {0, 0, 0, 0, 0};
            for (long k0 = 0; k0 < ndw; k0 += 4) {
                Raw r;
                for (int i = 0; i < 6; i++) { r.x[i] = host_dword(row, k0 - 2 + i, n); r.u[i] = host_dword(prev, k0 - 2 + i, n); }
                if (4 * (k0 + 4) <= n) { if (strategy == S_ADAPTIVE_FAST) score_group<BPP, false, true>(r, (int)k0, (int)n, sc); else score_group<BPP, false, false>(r, (int)k0, (int)n, sc); }
                else { if (strategy == S_ADAPTIVE_FAST) score_group<BPP, true, true>(r, (int)k0, (int)n, sc); else score_group<BPP, true, false>(r, (int)k0, (int)n, sc); }
            }
            unsigned long long tot[5];
            for (int i = 0; i < 5; i++) tot[i] = sc[i];
            f = decide(strategy, tot, (unsigned long long)n);
        }
        uint8_t *o = out + y * (n + 1);
        o[0] = (uint8_t)f;
        const uint64_t L = (uint64_t)n + 1;
        uint64_t s1 = (unsigned)f, s2 = L * (unsigned)f;
        for (long k0 = 0; k0 < ndw; k0 += 4) {
            Raw r;
            for (int i = 0; i < 6; i++) { r.x[i] = host_dword(row, k0 - 2 + i, n); r.u[i] = host_dword(prev, k0 - 2 + i, n); }
            Group g;
            group_of<BPP, true>(r, (int)k0, (int)n, g);
            uint32_t v[4], sum, ramp;
            for (int j = 0; j < 4; j++) {
                v[j] = filtered(f, g, j) & g.valid[j];
                for (int b = 0; b < 4; b++)
                    if (4 * (k0 + j) + b < n) o[1 + 4 * (k0 + j) + b] = (uint8_t)(v[j] >> (8 * b));
            }
            adler_terms16(v, sum, ramp); // (the kernel's emit_group: one multiplication per group, modular for the last one)
            s1 += sum;
            s2 += (L - (uint64_t)(4 * k0 + 16)) * sum + ramp;
        }
        // combine_adler (png_api.cpp): s2 += row_len * s1_before + B, s1 += A
        a2 = (a2 + (L % M) * a1 + s2 % M) % M;
        a1 = (a1 + s1 % M) % M;
    }
    return (uint32_t)((a2 << 16) | a1);
}
} // namespace
extern "C" long emu_png_filter(const uint8_t *data, long width, long height, int bpp, int strategy, uint8_t *out)
{
    const long n = width * bpp;
    switch (bpp) {
    case 1: return emu_png_rows<1>(data, n, height, strategy, out);
    case 2: return emu_png_rows<2>(data, n, height, strategy, out);
    case 3: return emu_png_rows<3>(data, n, height, strategy, out);
    case 4: return emu_png_rows<4>(data, n, height, strategy, out);
    case 6: return emu_png_rows<6>(data, n, height, strategy, out);
    case 8: return emu_png_rows<8>(data, n, height, strategy, out);
    default: return -1;
    }
}

// ---------------------------------------------------------------------------------------------
// The INTEGER secondary mode (pixo_amd/csrc/jpeg_int_math.h) compiled for the host.
// ---------------------------------------------------------------------------------------------
#include "../../pixo_amd/csrc/jpeg_int_math.h"
extern "C" void emu_int_dct_fast(const int16_t *block, int32_t *out)
{
    for (int i = 0; i < 64; i++) out[i] = block[i];
    pixo_int::dct_2d_fast(out);
}
extern "C" void emu_int_quantize(const int32_t *dct, const uint16_t *q, int16_t *out)
{
    for (int i = 0; i < 64; i++) out[i] = (int16_t)pixo_int::quantize_integer(dct[i], q[i]);
}
extern "C" void emu_int_color(int r, int g, int b, int32_t *out)
{
    const pixo_int::YCbCr c = pixo_int::rgb_to_ycbcr_2p16(r, g, b);
    out[0] = c.y; out[1] = c.cb; out[2] = c.cr;
}

// The single-walk form of jpeg_scan_fused.hip's code kernel, one group at a time on the CPU: every lane codes its block
// into a private scratch of `scratch_words` words (words beyond it are dropped, like on the device), the lengths are
// prefix-summed, and — unless a block was longer than the scratch — the scratches are OR-ed, shifted, into the group's
// zeroed bit buffer in windows of `window_words`.  Returns the packed (unstuffed, unpadded) bits of the whole scan as
// MSB-first words, or -1 when some group holds a long block (the device then takes its two-walk path).
extern "C" long emu_scan_single_walk(const int16_t *y, const int16_t *cb, const int16_t *cr, int mode, uint64_t nblocks,
                                     const uint32_t *tables, uint32_t scratch_words, uint32_t window_words, uint32_t *out_words,
                                     uint64_t *total_bits)
{
    using namespace pixo_scan;
    const int kGroupLanes = 192;
    uint32_t wtab[kWalkWords];
    for (int i = 0; i < kWalkWords; i++) wtab[i] = walk_table_word(tables, i);
    std::vector<uint32_t> stream;
    uint64_t before = 0;
    long long_groups = 0;
    for (uint64_t g0 = 0; g0 < nblocks; g0 += kGroupLanes) {
        const int lanes = (int)std::min<uint64_t>(kGroupLanes, nblocks - g0);
        std::vector<std::vector<uint32_t>> scratch(lanes, std::vector<uint32_t>(scratch_words + 1, 0xDEADBEEFu));
        std::vector<uint32_t> len(lanes), my_bit(lanes);
        uint32_t group_bits = 0;
        bool group_long = false;
        for (int l = 0; l < lanes; l++) {
            const BlockRef r = block_of(mode, g0 + l);
            const int16_t *base = r.comp == 0 ? y : (r.comp == 1 ? cb : cr);
            uint32_t wds[32];
            memcpy(wds, base + r.index * 64, 128);
            const int prev = r.index ? base[(r.index - 1) * 64] : 0;
            FlatPack<EmuLaneSink> p;
            p.sink = EmuLaneSink{scratch[l].data(), scratch_words};
            p.acc = 0; p.pending = 0; p.word = 0;
            block_pack_flat(wds, prev, wtab + (r.comp ? 1 : 0) * kWalkClassWords, p);
            len[l] = p.word * 32u + p.pending;
            p.finish();
            my_bit[l] = group_bits;
            group_bits += len[l];
            group_long |= len[l] > scratch_words * 32u;
        }
        if (group_long) { long_groups++; }
        const uint32_t local_words = (group_bits + 31) >> 5;
        std::vector<uint32_t> local(local_words + 1, 0);
        if (!group_long) {
            for (uint32_t wbase = 0; wbase < local_words; wbase += window_words) {
                const uint32_t wn = std::min(local_words - wbase, window_words);
                std::vector<uint32_t> buf(window_words + 1, 0); // [window_words] = dummy
                for (int l = 0; l < lanes; l++) {
                    const int64_t rel = (int64_t)my_bit[l] - (int64_t)wbase * 32;
                    const uint32_t nw = (len[l] + 31) >> 5, bsh = (uint32_t)(rel & 31), d0 = (uint32_t)(rel >> 5);
                    for (uint32_t j = 0; j < scratch_words; j++) {
                        if (j >= nw) break;
                        const uint32_t v = scratch[l][j], d = d0 + j;
                        buf[d < wn ? d : window_words] |= v >> bsh;
                        buf[d + 1 < wn ? d + 1 : window_words] |= bsh ? v << (32 - bsh) : 0u;
                    }
                }
                for (uint32_t i = 0; i < wn; i++) local[wbase + i] = buf[i];
            }
        } else { // the device's second walk: block_pack_flat through a windowed OR sink, round by round
            struct WinSink {
                uint32_t *buf; uint32_t limit, dummy;
                void or_word(bool flush, uint32_t word, uint32_t value) { buf[(flush && word < limit) ? word : dummy] |= value; }
            };
            for (uint32_t wbase = 0; wbase < local_words; wbase += window_words) {
                const uint32_t wn = std::min(local_words - wbase, window_words);
                std::vector<uint32_t> buf(window_words + 1, 0);
                for (int l = 0; l < lanes; l++) {
                    const int64_t rel = (int64_t)my_bit[l] - (int64_t)wbase * 32;
                    if (!(rel < (int64_t)wn * 32 && rel + (int64_t)len[l] > 0)) continue;
                    const BlockRef r = block_of(mode, g0 + l);
                    const int16_t *base = r.comp == 0 ? y : (r.comp == 1 ? cb : cr);
                    uint32_t wds[32];
                    memcpy(wds, base + r.index * 64, 128);
                    FlatPack<WinSink> p;
                    p.sink = WinSink{buf.data(), wn, window_words};
                    p.acc = 0; p.pending = (uint32_t)(rel & 31); p.word = (uint32_t)(rel >> 5);
                    block_pack_flat(wds, r.index ? base[(r.index - 1) * 64] : 0, wtab + (r.comp ? 1 : 0) * kWalkClassWords, p);
                    p.finish();
                }
                for (uint32_t i = 0; i < wn; i++) local[wbase + i] = buf[i];
            }
        }
        // append the group's bits at `before`
        stream.resize((before + group_bits + 63) / 32 + 2, 0);
        const uint32_t sh = (uint32_t)(before & 31);
        for (uint32_t i = 0; i < local_words; i++) {
            stream[(before >> 5) + i] |= local[i] >> sh;
            if (sh) stream[(before >> 5) + i + 1] |= local[i] << (32 - sh);
        }
        before += group_bits;
    }
    *total_bits = before;
    for (uint64_t i = 0; i < (before + 31) / 32; i++) out_words[i] = stream[i];
    return long_groups;
}
