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

template <int MODE, bool FAST>
static void run_image(const TileCtx &c, uint32_t tiles_x, uint32_t tiles_y, long *stats, int wave_order)
{
    typedef Geo<MODE> G;
    static_assert(lds_bytes<MODE>() <= 160 * 1024 / 6, "six workgroups must fit a CU's LDS");
    alignas(16) static uint8_t lds[lds_bytes<MODE>()];
    std::vector<float> v((size_t)192 * 64);
    uint32_t regs[G::items * G::item_regs];
    for (uint32_t ty = 0; ty < tiles_y; ty++)
        for (uint32_t tx = 0; tx < tiles_x; tx++) {
            if (stats) stats[FAST ? 0 : 1]++; // tiles read by vector loads / by byte gathers
            uint8_t *planar = lds;
            memset(planar, 0xA5, G::planar); // nothing may rely on a previous tile's samples
            // producer wavefront: every lane loads and converts its items of this tile
            for (int lane = 0; lane < 64; lane++) {
                for (int k = 0; k < G::items; k++) producer_load_item<MODE, FAST>(c, tx, ty, k, lane, &regs[k * G::item_regs]);
                for (int k = 0; k < G::items; k++) {
                    producer_fix_item<MODE, FAST>(c, tx, k, lane, &regs[k * G::item_regs]);
                    producer_color_item<MODE>(k, lane, &regs[k * G::item_regs], planar);
                }
            }
            // (barrier) consumer wavefronts, in a caller-chosen order: they share nothing
            for (int k = 0; k < 3; k++) {
                const int w = wave_order == 0 ? k : (wave_order == 1 ? 2 - k : (k + 1) % 3);
                // a wavefront runs these steps in lockstep: every lane finishes a step before any
                // lane starts the next (the stage is written by block and read back by chunk)
                for (int l = 0; l < 64; l++) consumer_rows<MODE>(w, l, planar, &v[(w * 64 + l) * 64]);
                for (int l = 0; l < 64; l++) consumer_cols(&v[(w * 64 + l) * 64]);
                for (int half = 0; half < 2; half++) {
                    for (int l = 0; l < 64; l++) consumer_quant_half<MODE>(w, l, c.qt, &v[(w * 64 + l) * 64], half, lds + stage_offset<MODE>(w));
                    for (int l = 0; l < 64; l++) consumer_store_half<MODE>(c, tx, ty, w, l, half, lds + stage_offset<MODE>(w));
                }
            }
        }
}

extern "C" int emu_jpeg_coeffs(const uint8_t *px, uint32_t W, uint32_t H, int color_type, int subsampling,
                               int quality, int16_t *y, int16_t *cb, int16_t *cr, int allow_fast,
                               long *stats /* [interior tiles, edge tiles] or NULL */, int wave_order)
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
    c.fast = allow_fast && ((uintptr_t)px % 4 == 0) && (row_bytes % 4 == 0) && W >= 4;
    if (stats) stats[0] = stats[1] = 0;
    if (gray) {
        if (c.fast) run_image<MGRAY, true>(c, (c.units_x + 63) / 64, (c.units_y + 2) / 3, stats, wave_order);
        else run_image<MGRAY, false>(c, (c.units_x + 63) / 64, (c.units_y + 2) / 3, stats, wave_order);
    } else if (s420) {
        if (c.fast) run_image<M420, true>(c, (c.units_x + 31) / 32, c.units_y, stats, wave_order);
        else run_image<M420, false>(c, (c.units_x + 31) / 32, c.units_y, stats, wave_order);
    } else {
        if (c.fast) run_image<M444, true>(c, (c.units_x + 63) / 64, c.units_y, stats, wave_order);
        else run_image<M444, false>(c, (c.units_x + 63) / 64, c.units_y, stats, wave_order);
    }
    return 0;
}

// quantiser fast path vs the reference formula on caller-provided values (for the
// dense/exhaustive sweeps in tests/test_quant_exact.py)
extern "C" long emu_quant_mismatches(const float *x, long n, int qlo, int qhi)
{
    long bad = 0;
    for (int q = qlo; q <= qhi; q++) {
        const float fq = (float)q, rcp = 1.0f / fq;
        float rr[8], qq[8], xx[8];
        for (int i = 0; i < 8; i++) { rr[i] = rcp; qq[i] = fq; }
        for (long i = 0; i + 8 <= n; i += 8) {
            uint32_t out[4];
            for (int k = 0; k < 8; k++) xx[k] = x[i + k];
            quant_row8(xx, rr, as_qtab(qq), 1.0f, out);
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
    const float fq = (float)q, rcp = 1.0f / fq;
    long f = 0, w = 0;
    for (long i = 0; i < n; i++) {
        float r = x[i] * rcp;
        float sm = r + kRoundMagic;
        bool risky = quant_risk(r, sm) >= 0.5f; // the kernel's own test (jpeg_tile.h)
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
    const float fq = (float)q, rcp = 1.0f / fq;
    long f = 0, w = 0, wf = 0;
    for (int sign = 0; sign < 2; sign++)
        for (uint64_t b = lo_bits; b <= hi_bits; b++) {
            uint32_t u = (uint32_t)b | (sign ? 0x80000000u : 0u);
            float x;
            memcpy(&x, &u, 4);
            float want = roundf(x / fq);
            float r = x * rcp;
            float sm = r + kRoundMagic;
            bool risky = quant_risk(r, sm) >= 0.5f; // the kernel's own test (jpeg_tile.h)
            float got = (float)(int16_t)(__builtin_bit_cast(uint32_t, sm) & 0xFFFF);
            if (risky) { f++; got = want; }
            else if (got != want) w++;
            if (got != want) wf++;
        }
    *flagged = f; *wrong_unflagged = w; *wrong_final = wf;
}
