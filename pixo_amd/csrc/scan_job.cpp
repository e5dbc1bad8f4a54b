// scan_job.cpp — coefficient launches on a context, and ONE pass of the device entropy stage over a coefficient tuple in
// HBM in the steps a caller may need to interleave with exchanges (whole image, batch of images, band of a larger image).
#include <algorithm>
#include <atomic>

#include "capi_internal.hpp"

namespace pixo_capi {

// Runs the device pipeline for host pixels; on success `*coef` points at pinned host
// memory holding [y | cb | cr] contiguously.
int coeffs_to_pinned(Context &c, const uint8_t *pixels, const pixo_jpeg_options &o, const pixo_host::Geometry &g,
                     const int16_t **y, const int16_t **cb, const int16_t **cr)
{
    int rc = c.ensure();
    if (rc) return rc;
    PIXO_ON_DEVICE_OF(c);
    const float *qt_all = nullptr;
    rc = device_tables(c.device, &qt_all);
    if (rc) return rc;
    const size_t px_bytes = static_cast<size_t>(o.width) * o.height * (g.gray ? 1 : 3);
    const size_t coef_bytes = (g.y_blocks + 2 * g.c_blocks) * 128;
    if ((rc = c.reserve_px((px_bytes + 15) & ~size_t{15}))) return rc;
    if ((rc = c.reserve_coef(coef_bytes))) return rc;
    if ((rc = c.reserve_hcoef(coef_bytes))) return rc;
    HIP_TRY(hipMemcpyAsync(c.d_px, pixels, px_bytes, hipMemcpyHostToDevice, c.stream));
    int16_t *dy = static_cast<int16_t *>(c.d_coef);
    int16_t *dcb = dy + g.y_blocks * 64;
    int16_t *dcr = dcb + g.c_blocks * 64;
    HIP_TRY(pixo_dev::launch_jpeg_coeffs(c.d_px, o.width, o.height, g.gray, g.s420, 1, dy,
                                         g.gray ? nullptr : dcb, g.gray ? nullptr : dcr,
                                         qt_all + (o.quality - 1) * pixo_host::kDeviceQtFloats, c.stream));
    HIP_TRY(hipMemcpyAsync(c.h_coef, c.d_coef, coef_bytes, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    *y = static_cast<const int16_t *>(c.h_coef);
    *cb = *y + g.y_blocks * 64;
    *cr = *cb + g.c_blocks * 64;
    return PIXO_OK;
}

// Device pixels -> device coefficient tuple inside the context's buffer.
// The tuple's place in the context (no launch)
int coeffs_reserve(Context &c, const pixo_host::Geometry &g, int16_t **dy, int16_t **dcb, int16_t **dcr)
{
    const size_t coef_bytes = (g.y_blocks + 2 * g.c_blocks) * 128;
    const int rc = c.reserve_coef(coef_bytes);
    if (rc) return rc;
    *dy = static_cast<int16_t *>(c.d_coef);
    *dcb = *dy + g.y_blocks * 64;
    *dcr = *dcb + g.c_blocks * 64;
    return PIXO_OK;
}
// The coefficient kernel over MCU rows [row0, row0 + rows) of the image — a sub-image of the same width whose blocks
// land at their places in the whole image's tuple (MCU rows are independent: SURVEY §8e; the last rows replicate the
// image's bottom edge as the whole-image launch does).  rows = 0: to the end.
int coeffs_rows(Context &c, const void *d_pixels, const pixo_jpeg_options &o, const pixo_host::Geometry &g, hipStream_t stream,
                int16_t *dy, int16_t *dcb, int16_t *dcr, uint32_t row0, uint32_t rows)
{
    const float *qt_all = nullptr;
    const int rc = device_tables(c.device, &qt_all);
    if (rc) return rc;
    const uint32_t unit = (!g.gray && g.s420) ? 16u : 8u, units_x = (o.width + unit - 1) / unit, units_y = (o.height + unit - 1) / unit;
    if (rows == 0 || row0 + rows > units_y) rows = units_y - row0;
    const uint32_t y0 = row0 * unit, y1 = row0 + rows >= units_y ? o.height : (row0 + rows) * unit;
    const size_t bpp = g.gray ? 1 : 3, m0 = static_cast<size_t>(row0) * units_x;
    const uint8_t *px = static_cast<const uint8_t *>(d_pixels) + static_cast<size_t>(y0) * o.width * bpp;
    HIP_TRY(pixo_dev::launch_jpeg_coeffs(px, o.width, y1 - y0, g.gray, g.s420, 1, dy + m0 * (unit == 16 ? 4 : 1) * 64,
                                         g.gray ? nullptr : dcb + m0 * 64, g.gray ? nullptr : dcr + m0 * 64,
                                         qt_all + (o.quality - 1) * pixo_host::kDeviceQtFloats, stream));
    return PIXO_OK;
}
int coeffs_on_device(Context &c, const void *d_pixels, const pixo_jpeg_options &o, const pixo_host::Geometry &g, hipStream_t stream,
                     int16_t **dy, int16_t **dcb, int16_t **dcr)
{
    const int rc = coeffs_reserve(c, g, dy, dcb, dcr);
    if (rc) return rc;
    return coeffs_rows(c, d_pixels, o, g, stream, *dy, *dcb, *dcr, 0, 0);
}

// Does the scan emit RSTn markers (jpeg/mod.rs:1431-1445: only while more MCUs follow)?
bool scan_has_restart_markers(const pixo_jpeg_options &o, const pixo_host::Geometry &g)
{
    return o.has_restart_interval && o.restart_interval != 0 && o.restart_interval < g.units;
}

// ---- the device entropy stage, in the steps a caller may need to interleave with exchanges ------------
// One pass of jpeg_entropy.hip over a coefficient tuple in HBM: a whole image, a batch of images (one
// byte-aligned segment each), or a BAND of a larger image (predictors seeded from the band
// above, packed at the band's bit offset modulo 8, no final padding).

// The packed Huffman tables of a scan into e_tables — unless they are what the buffer holds already (the standard
// tables, image after image: one small copy less on the stream per file).
int upload_scan_tables(Context &c, const uint32_t (&packed)[pixo_host::kScanTableWords], hipStream_t stream)
{
    if (c.tables_valid && c.tables_stream == stream && std::memcmp(c.tables_held, packed, sizeof packed) == 0) return PIXO_OK;
    c.tables_valid = false;
    std::memcpy(c.tables_held, packed, sizeof packed);
    for (int i = 0; i < pixo_scan::kWalkWords; ++i) // the same tables in the form of the flat walk (jpeg_scan_block.h)
        c.tables_held[pixo_scan::kTableWords + i] = pixo_scan::walk_table_word(packed, i);
    // (from pinned words: the copy is a plain DMA, not the runtime's staging of pageable memory.  The staging words are free again:
    // every entry point ends with the stream synchronised, and a call uploads its tables once per stream)
    if (!c.h_tables) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c.h_tables), sizeof c.tables_held, hipHostMallocDefault));
    std::memcpy(c.h_tables, c.tables_held, sizeof c.tables_held);
    HIP_TRY(hipMemcpyAsync(c.e_tables.p, c.h_tables, sizeof c.tables_held, hipMemcpyHostToDevice, stream));
    c.tables_valid = true;
    c.tables_stream = stream;
    return PIXO_OK;
}

// The single-pass kernels bound their waits (jpeg_scan_fused.hip): a launch that gave up says so in the pinned mailbox
// (h_totals[3]); its outputs are garbage.  The caller of the step that notices gets kRetryMultipass back and codes the scan
// again with t_force_multipass set (RetryMultipass below): the multi-pass kernels wait for nothing but kernel boundaries.
thread_local bool t_force_multipass = false;
std::atomic<uint64_t> g_lookback_fallbacks{0};
int scan_retry_multipass(Context &c)
{
    c.code_state_zero_words = 0; // (the descriptors and the flag are dirty: the next single-pass launch starts with a memset)
    g_lookback_fallbacks.fetch_add(1, std::memory_order_relaxed);
    return kRetryMultipass;
}
uint64_t lookback_fallbacks() { return g_lookback_fallbacks.load(std::memory_order_relaxed); }

constexpr uint64_t kMinSegBlocks = 96; // a segment's groups are aligned with it: at least half a group of 192 blocks

// Geometry of the pass and every buffer whose size does not depend on the data.
int scan_begin(Context &c, ScanJob &j, const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
               const pixo_host::Geometry &g, uint32_t batch, const int16_t *band_seed_dc)
{
    namespace pd = pixo_dev;
    // every pass starts from scratch: a job object that is kept across calls (the band encoder's) must not carry the last
    // image's tables, lengths or flags into the next one (scan_tables would otherwise return early with the old tables)
    const uint32_t seg_gap = j.seg_gap; // (the one input a caller sets before the pass begins)
    j = ScanJob{};
    j.seg_gap = seg_gap <= pd::seg_max_gap() ? seg_gap : 0;
    j.n = (g.y_blocks + 2 * g.c_blocks) * batch;
    pd::ScanArgs &a = j.a;
    a.y = dy; a.cb = dcb; a.cr = dcr;
    a.mode = g.gray ? 0 : (g.s420 ? 2 : 1);
    a.nblocks = j.n;
    a.blocks_per_mcu = g.gray ? 1 : (g.s420 ? 6 : 3);
    a.marker_bytes = 2;
    a.restart = (!band_seed_dc && scan_has_restart_markers(o, g)) ? o.restart_interval : 0;
    a.seed_dc[0] = a.seed_dc[1] = a.seed_dc[2] = 0;
    a.bit_base = 0; a.pad_last = 1;
    j.band = band_seed_dc != nullptr;
    if (j.band) {
        for (int i = 0; i < 3; ++i) a.seed_dc[i] = band_seed_dc[i];
        a.pad_last = 0;
    }
    j.nseg = a.restart ? (g.units + a.restart - 1) / a.restart : 0;
    if (batch > 1) { // one segment per image, no marker between them
        a.restart = static_cast<uint32_t>(g.units);
        a.marker_bytes = 0;
        j.nseg = batch;
    }
    // Which kernels: the two single-pass kernels of jpeg_scan_fused.hip for an uninterrupted scan and — since round 3 — for
    // segments of at least kMinSegBlocks blocks (the images of a batch; restart intervals of 16 MCUs of 4:2:0 or more:
    // a segment's groups of 192 blocks are aligned with it, so short segments would leave most lanes idle); the multi-pass
    // kernels of jpeg_entropy.hip for short restart intervals, on request (debug switch), and after a single-pass
    // launch gave up waiting (t_force_multipass, see scan_retry_multipass).
    const uint64_t seg_blocks = static_cast<uint64_t>(a.restart) * a.blocks_per_mcu;
    const bool single_pass = !debug().multipass_entropy && !t_force_multipass;
    j.segmented = single_pass && j.nseg > 0 && seg_blocks >= kMinSegBlocks && seg_blocks <= 0xFFFFFFFFull;
    j.fused = single_pass && (j.nseg == 0 || j.segmented);
    HIP_TRY(c.e_tables.reserve(pixo_scan::kScanTableUpload * 4));
    HIP_TRY(c.e_hist.reserve(pixo_host::kScanTableWords * 8));
    if (j.segmented) {
        pd::SegArgs &sg = j.seg;
        sg.nsegs = j.nseg;
        sg.blocks = static_cast<uint32_t>(seg_blocks);
        sg.groups = pd::seg_groups(seg_blocks);
        sg.stream_words = (seg_blocks * 209 + 64 + 15) / 16 * 4;
        sg.marker_bytes = a.marker_bytes ? 2u : j.seg_gap; // RSTn, or the gap a batch wants between its files' scans
        sg.rst_markers = a.marker_bytes ? 1u : 0u;
        j.stream_cap = static_cast<size_t>(sg.stream_words) * 4 * j.nseg;
        HIP_TRY(c.e_stream.reserve(j.stream_cap + 64));
        const size_t state_words = pd::fused_code_state_words_seg(j.nseg, seg_blocks);
        if (state_words * 8 > c.e_code_state.cap) c.code_state_zero_words = 0; // (a new buffer)
        HIP_TRY(c.e_code_state.reserve(state_words * 8));
        j.code_state_words = state_words;
        // every segment's last tile is partial: one tile more per segment than the bytes alone would need
        HIP_TRY(c.e_stuff_state.reserve((pd::fused_stuff_state_words(j.stream_cap) + j.nseg) * 8));
        HIP_TRY(c.e_segs.reserve((4 * j.nseg + 2) * 8));
        unsigned long long *base = c.e_segs.as<unsigned long long>();
        sg.bits = base;
        sg.layout = base + j.nseg;
        sg.bytes = base + 2 * j.nseg + 2;
        sg.out_end = base + 3 * j.nseg + 2;
        // (room for the optimised-tables counters as well: scan_count / pixels_count reserve theirs AFTER this address has been handed out)
        { const int rc_s = c.reserve_hsegs(std::max<size_t>(j.nseg, pixo_host::kScanTableWords)); if (rc_s) return rc_s; }
        sg.host_out_end = reinterpret_cast<unsigned long long *>(c.h_segs);
        { const int rc_t = c.ensure_totals(); if (rc_t) return rc_t; }
        a.tables = c.e_tables.as<uint32_t>();
        a.pad_last = 1;
        return PIXO_OK;
    }
    if (j.fused) { // a block has at most 1665 bits: the packed stream has at most n * 209 bytes (+ slack the kernels read into)
        j.stream_cap = static_cast<size_t>(j.n) * 209 + 64;
        HIP_TRY(c.e_stream.reserve(j.stream_cap));
        if (pd::fused_code_state_words(j.n) * 8 > c.e_code_state.cap) c.code_state_zero_words = 0; // (a new buffer)
        HIP_TRY(c.e_code_state.reserve(pd::fused_code_state_words(j.n) * 8));
        j.code_state_words = pd::fused_code_state_words(j.n);
        HIP_TRY(c.e_stuff_state.reserve(pd::fused_stuff_state_words(j.stream_cap) * 8));
        { const int rc_t = c.ensure_totals(); if (rc_t) return rc_t; }
        a.tables = c.e_tables.as<uint32_t>();
        return PIXO_OK;
    }
    HIP_TRY(c.e_len.reserve((j.n ? j.n : 1) * 4));
    HIP_TRY(c.e_off.reserve((j.n ? j.n : 1) * 8));
    // scratch of the three prefix sums (blocks, restart segments, 0xFF tiles), reserved before any launch:
    // a block has at most 1665 bits, so the packed stream has at most n * 209 + 3 * nseg bytes
    j.tmp_blocks = pd::scan_tile_count(j.n) + 1; j.tmp_segs = pd::scan_tile_count(j.nseg ? j.nseg : 1) + 1;
    j.tmp_tiles = pd::scan_tile_count(pd::stuff_tile_count(j.n * 209 + 3 * j.nseg + 8)) + 1;
    HIP_TRY(c.e_tmp.reserve((j.tmp_blocks + j.tmp_segs + j.tmp_tiles) * 8));
    HIP_TRY(c.e_totals.reserve(16));
    { const int rc_t = c.ensure_totals(); if (rc_t) return rc_t; }
    a.tables = c.e_tables.as<uint32_t>();
    return PIXO_OK;
}

void split_counts(const uint64_t counts[pixo_host::kScanTableWords], uint64_t dc[2][12], uint64_t ac[2][256])
{
    for (int cls = 0; cls < 2; ++cls) {
        std::memcpy(dc[cls], counts + cls * 268, sizeof dc[cls]);
        std::memcpy(ac[cls], counts + cls * 268 + 12, sizeof ac[cls]);
    }
}

// count_block statistics of the pass (src/jpeg/mod.rs:826-860) gathered on the device: [class][12 DC + 256 AC].
int scan_count(Context &c, ScanJob &j, hipStream_t stream, uint64_t counts[pixo_host::kScanTableWords])
{
    HIP_TRY(c.e_count.reserve(pixo_dev::scan_count_scratch_bytes()));
    HIP_TRY(pixo_dev::launch_scan_count(j.a, c.e_count.as<uint32_t>(), c.e_hist.as<unsigned long long>(), stream));
    // (into the context's pinned words and from there to the caller's array: a copy to pageable memory goes through the runtime's
    // staging buffer and costs a small optimised-tables file about ten microseconds)
    { const int rc = c.reserve_hsegs(pixo_host::kScanTableWords); if (rc) return rc; }
    HIP_TRY(hipMemcpyAsync(c.h_segs, c.e_hist.p, pixo_host::kScanTableWords * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    std::memcpy(counts, c.h_segs, pixo_host::kScanTableWords * 8);
    return PIXO_OK;
}

// The same statistics straight from the pixels (launch_pixels_count: the fused kernel's scans never have a tuple).
int pixels_count(Context &c, const pixo_jpeg_options &o, const pixo_host::Geometry &g, hipStream_t stream, const void *d_pixels,
                 uint64_t counts[pixo_host::kScanTableWords])
{
    namespace pd = pixo_dev;
    const float *qt_all = nullptr;
    { const int rc = device_tables(c.device, &qt_all); if (rc) return rc; }
    const uint32_t restart = scan_has_restart_markers(o, g) ? o.restart_interval : 0;
    const pd::PixelsCodePlan plan = pd::pixels_code_plan(o.width, o.height, g.s420, 1, restart, g.gray);
    HIP_TRY(c.e_count.reserve(pd::pixels_count_scratch_bytes(plan)));
    HIP_TRY(pd::launch_pixels_count(d_pixels, o.width, o.height, g.gray, g.s420, plan, qt_all + (o.quality - 1) * pixo_host::kDeviceQtFloats, c.e_count.p,
                                    c.e_hist.as<unsigned long long>(), stream));
    { const int rc = c.reserve_hsegs(pixo_host::kScanTableWords); if (rc) return rc; }
    HIP_TRY(hipMemcpyAsync(c.h_segs, c.e_hist.p, pixo_host::kScanTableWords * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    std::memcpy(counts, c.h_segs, pixo_host::kScanTableWords * 8);
    return PIXO_OK;
}

// Tables (standard; optimised from `counts`, or from this pass's own statistics when counts == null),
// block bit lengths and their prefix sum: afterwards j.total_bits is known (one read-back).
int scan_tables(Context &c, ScanJob &j, const pixo_jpeg_options &o, const pixo_host::Geometry &g, hipStream_t stream,
                const uint64_t *counts)
{
    if (j.tables_ready) return PIXO_OK;
    if (o.optimize_huffman) { // table construction on the host, exactly like optimized_from_counts
        uint64_t own[pixo_host::kScanTableWords];
        if (!counts) {
            int rc = j.count_px ? pixels_count(c, o, g, stream, j.count_px, own) : scan_count(c, j, stream, own);
            if (rc) return rc;
            counts = own;
        }
        uint64_t dc[2][12], ac[2][256];
        split_counts(counts, dc, ac);
        j.h = pixo_host::HuffSet::optimized(dc, ac, !g.gray);
    } else {
        j.h = pixo_host::HuffSet::standard();
    }
    uint32_t packed[pixo_host::kScanTableWords];
    pixo_host::pack_scan_tables(j.h, packed);
    const int rc = upload_scan_tables(c, packed, stream);
    j.tables_ready = rc == PIXO_OK;
    return rc;
}

int scan_lengths(Context &c, ScanJob &j, const pixo_jpeg_options &o, const pixo_host::Geometry &g, hipStream_t stream,
                 const uint64_t *counts, bool wait)
{
    namespace pd = pixo_dev;
    {
        const int rc = scan_tables(c, j, o, g, stream, counts);
        if (rc) return rc;
    }
    if (j.segmented) { // every segment packed into its own stream from bit 0; always chained with the stuffing kernel
        const size_t state_words = pd::fused_code_state_words_seg(j.seg.nsegs, j.seg.blocks);
        const bool zero = c.code_state_zero_words >= state_words;
        c.code_state_zero_words = 0;
        HIP_TRY(pd::launch_scan_code(j.a, c.e_code_state.as<unsigned long long>(), zero, c.e_stream.as<uint32_t>(),
                                     c.e_stuff_state.as<unsigned long long>(), pd::fused_stuff_state_words(j.stream_cap) + j.nseg,
                                     reinterpret_cast<unsigned long long *>(c.h_totals), stream, nullptr, &j.seg, debug().spin_budget));
        HIP_TRY(pd::launch_seg_layout(j.seg, const_cast<unsigned long long *>(j.seg.layout), const_cast<unsigned long long *>(j.seg.bytes),
                                      reinterpret_cast<unsigned long long *>(c.h_totals), stream));
        if (wait) { // (a caller that wants the lengths now: none of the product's paths; kept for symmetry)
            HIP_TRY(hipStreamSynchronize(stream));
            if (c.h_totals[3]) return scan_retry_multipass(c);
        }
        return PIXO_OK;
    }
    if (j.fused) { // lengths, prefix and packing in one pass; the stream starts at bit 0 whatever the band's offset will be
        const bool zero = c.code_state_zero_words >= pd::fused_code_state_words(j.n);
        c.code_state_zero_words = 0; // (dirty from here until a stuffing launch has cleaned it)
        // chained with the stuffing kernel (!wait): this launch also zeroes that kernel's descriptors
        HIP_TRY(pd::launch_scan_code(j.a, c.e_code_state.as<unsigned long long>(), zero, c.e_stream.as<uint32_t>(),
                                     wait ? nullptr : c.e_stuff_state.as<unsigned long long>(),
                                     wait ? 0 : pd::fused_stuff_state_words(j.stream_cap), reinterpret_cast<unsigned long long *>(c.h_totals), stream,
                                     nullptr, nullptr, debug().spin_budget));
        if (!wait) return PIXO_OK; // (the caller chains the stuffing kernel and synchronises once)
        HIP_TRY(hipStreamSynchronize(stream)); // (the kernel wrote the length into the pinned mailbox itself)
        if (c.h_totals[3]) return scan_retry_multipass(c);
        j.total_bits = c.h_totals[0];
        j.nbytes = (j.total_bits + 7) / 8;
        return PIXO_OK;
    }
    if (j.n) HIP_TRY(pd::launch_scan_lengths(j.a, c.e_len.as<uint32_t>(), stream));
    HIP_TRY(pd::launch_exclusive_scan(c.e_len.as<uint32_t>(), j.n, c.e_off.as<uint64_t>(), c.e_tmp.as<uint64_t>(),
                                      c.e_totals.as<uint64_t>(), stream));
    if (j.nseg) { // restart markers: byte-aligned segments, each followed by two marker bytes
        HIP_TRY(c.e_seg_bytes.reserve(j.nseg * 8));
        HIP_TRY(c.e_seg_off.reserve(j.nseg * 8));
        HIP_TRY(pd::launch_segment_sizes(j.a, c.e_off.as<uint64_t>(), c.e_totals.as<uint64_t>(), j.nseg, c.e_seg_bytes.as<uint32_t>(), stream));
        HIP_TRY(pd::launch_exclusive_scan(c.e_seg_bytes.as<uint32_t>(), j.nseg, c.e_seg_off.as<uint64_t>(),
                                          c.e_tmp.as<uint64_t>() + j.tmp_blocks, c.e_totals.as<uint64_t>() + 1, stream));
        j.plan.nsegments = j.nseg;
        j.plan.seg_byte_off = c.e_seg_off.as<uint64_t>();
    }
    HIP_TRY(hipMemcpyAsync(c.h_totals, c.e_totals.p, 16, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream)); // `packed` may go out of scope after this, too
    j.total_bits = c.h_totals[0];
    j.nbytes = j.nseg ? c.h_totals[1] : (j.total_bits + 7) / 8; // bytes of the packed (unstuffed) stream
    return PIXO_OK;
}

bool pixels_code_usable(const ScanJob &j, const pixo_jpeg_options &o, const pixo_host::Geometry &g, uint32_t batch)
{ // one uninterrupted RGB scan, the images of a batch, or restart intervals of whole MCU rows — with GIVEN tables.  (Independent of
  // which tuple kernels scan_begin chose: segments of any size are chains of the fused kernel.)
    if (j.band || o.progressive || debug().two_kernel_scan || debug().multipass_entropy || t_force_multipass) return false;
    if (o.optimize_huffman && batch > 1) return false; // (every file of a batch has tables of its own: not segments of one launch)
    // Batches: every image a segment of ONE launch of the fused kernel — since the segments' byte counts are asked for BEHIND a group's own
    // 0xFF count (jpeg_pixels_code.hip; in front of it every segment's last group finished 6.5 us behind the one before: 64 x 1080p took
    // 414-521 us) a batch runs at the rate of one large image: 64 x 1080p 245 / 272 / 340 us (gradient / photo / noise) against 306 / 335 /
    // 444 us through coefficient kernel + scan_code + stuffing kernel (profiles/r06_fused_batches_chain.txt).  The fused kernel's group is a
    // 512-pixel TILE: images whose tiles are mostly empty (640 px wide: 62 %) keep the two-kernel form, whose groups are dense
    // (256 x 640x480: 207-233 against 195-211 us).  debug switch fused_batch: the fused kernel whatever the width.
    if (batch > 1 && !debug().fused_batch) {
        const uint32_t unit = g.s420 ? 16u : 8u, per_tile = g.gray ? 192u : (g.s420 ? 32u : 64u);
        const uint32_t units_x = (o.width + unit - 1) / unit, tiles_x = (units_x + per_tile - 1) / per_tile;
        if (static_cast<uint64_t>(units_x) * 4 < static_cast<uint64_t>(tiles_x) * per_tile * 3) return false;
    }
    const uint32_t restart = (batch == 1 && scan_has_restart_markers(o, g)) ? o.restart_interval : 0;
    return pixo_dev::pixels_code_supported(o.width, o.height, g.gray, g.s420, batch, restart);
}

int scan_from_pixels(Context &c, ScanJob &j, const pixo_jpeg_options &o, const pixo_host::Geometry &g, hipStream_t stream, const void *d_pixels,
                     HostTarget *host, bool wait, uint32_t batch)
{
    namespace pd = pixo_dev;
    if (o.optimize_huffman && !j.tables_ready) j.count_px = d_pixels;
    int rc = scan_tables(c, j, o, g, stream, nullptr);
    if (rc) return rc;
    const float *qt_all = nullptr;
    if ((rc = device_tables(c.device, &qt_all))) return rc;
    const uint32_t restart = (batch == 1 && scan_has_restart_markers(o, g)) ? o.restart_interval : 0;
    const pd::PixelsCodePlan plan = pd::pixels_code_plan(o.width, o.height, g.s420, batch, restart, g.gray);
    const size_t words = plan.state_words;
    const bool segs = plan.segments > 1;
    const uint32_t gap = !segs ? 0u : (restart ? 2u : j.seg_gap); // RSTn, or what a batch wants between its files' scans
    const bool rst = segs && restart != 0;
    if (segs) { // (where every segment ends: the kernel's pinned mailbox, like the single-pass tuple kernels')
        if ((rc = c.reserve_hsegs(plan.segments))) return rc;
        j.pc_seg = true;
        j.seg.marker_bytes = gap;
    }
    // two state blocks: this launch's must be zero, and the launch zeroes the other one (the launch before it used that) on the side
    if (c.e_pc_state.cap < 2 * words * 8 || c.pc_half_words != words) {
        HIP_TRY(c.e_pc_state.reserve(2 * words * 8));
        HIP_TRY(hipMemsetAsync(c.e_pc_state.p, 0, 2 * words * 8, stream));
        c.pc_half_words = words;
        c.pc_flip = 0;
    }
    // groups of several 6 KiB rounds park their blocks here (jpeg_pixels_code.hip).  NOT in d_coef: the tuple pointers the caller
    // derived from it must stay valid for the multi-pass retry, and a hipFree would synchronise the device in the middle of the call
    HIP_TRY(c.e_pc_spill.reserve(static_cast<size_t>(plan.groups) * 192 * 128));
    size_t want_cap = std::max<size_t>(j.stream_cap / 4, 4096) + static_cast<size_t>(plan.segments) * gap;
    for (int attempt = 0;; ++attempt) {
        uint8_t *out = nullptr;
        size_t out_cap = 0;
        if (host) {
            if (host->grow) {
                const int rc_h = c.reserve_hfile(host->before + want_cap + host->after);
                if (rc_h) return rc_h;
                host->p = c.h_file + host->before;
                host->cap = c.hfile_cap - host->before - host->after;
            }
            out = host->p;
            out_cap = host->cap;
        } else {
            HIP_TRY(c.e_out.reserve(want_cap));
            out = c.e_out.as<uint8_t>();
            out_cap = c.e_out.cap;
        }
        unsigned long long *mine = c.e_pc_state.as<unsigned long long>() + static_cast<size_t>(c.pc_flip) * words;
        unsigned long long *other = c.e_pc_state.as<unsigned long long>() + static_cast<size_t>(c.pc_flip ^ 1) * words;
        c.pc_flip ^= 1;
        HIP_TRY(pd::launch_pixels_code(d_pixels, o.width, o.height, g.gray, g.s420, plan, gap, rst, qt_all + (o.quality - 1) * pixo_host::kDeviceQtFloats,
                                       c.e_tables.as<uint32_t>(), mine, /*state_is_zero=*/true, other, words, out, out_cap,
                                       reinterpret_cast<unsigned long long *>(c.h_totals), segs ? reinterpret_cast<unsigned long long *>(c.h_segs) : nullptr,
                                       nullptr, true, c.e_pc_spill.p, stream, debug().spin_budget));
        if (!wait) return PIXO_OK;
        HIP_TRY(hipStreamSynchronize(stream));
        if (c.h_totals[3]) { c.pc_half_words = 0; return scan_retry_multipass(c); } // (both blocks are memset before the next use)
        j.total_bits = c.h_totals[0];
        j.scan_bytes = c.h_totals[1];
        j.nbytes = segs ? 0 : c.h_totals[2];
        if (j.scan_bytes > out_cap) { // (nothing was stored beyond the capacity: more room, the same kernel again)
            if (host && !host->grow) return PIXO_OK; // (the caller's storage is what it is: the caller reports the size needed)
            if (attempt > 1) return fail(PIXO_ERR_COMPRESSION, "Compression error: scan larger than announced");
            want_cap = static_cast<size_t>(j.scan_bytes);
            continue;
        }
        return PIXO_OK;
    }
}

// The stuffing kernel of jpeg_scan_fused.hip over the packed stream (launch_scan_code has been enqueued; with
// `chained` its length has not been read back yet): afterwards c.e_out holds j.scan_bytes finished bytes.  The output
// buffer is sized from experience (grow-only) — the kernel never writes beyond it and says how much it needed.
// Where the stuffed bytes go when not into the context's device buffer: host memory the GPU can write (pinned), so that
// the kernel's stores ARE the transfer — no second pass over the file, no second synchronisation.

// Segmented scans: all segments' tiles in one launch (scan_lengths has enqueued code + layout).  Afterwards c.e_out holds
// j.scan_bytes bytes — the segments back to back, RSTn markers between them if the job has any — and c.h_segs[k] where
// segment k's bytes end.
int scan_stuff_segmented(Context &c, ScanJob &j, hipStream_t stream)
{
    namespace pd = pixo_dev;
    const size_t code_words = pd::fused_code_state_words_seg(j.seg.nsegs, j.seg.blocks);
    // tiles: a guess of 64 bytes per block (noise at q = 80 has 28) + one partial tile per segment; surplus workgroups
    // leave at once, missing ones are launched below once the layout kernel has said how many there are
    const uint64_t per_seg = pd::stuff_tiles(static_cast<uint64_t>(j.seg.blocks) * 64 + 4096);
    uint64_t first_tile = 0, tiles = per_seg * j.nseg;
    size_t want_cap = std::max<size_t>(j.stream_cap / 4, 4096);
    for (int attempt = 0;; ++attempt) {
        HIP_TRY(c.e_out.reserve(want_cap));
        HIP_TRY(pd::launch_stuff_fused(c.e_stream.as<uint32_t>(), c.e_code_state.as<unsigned long long>(), code_words, 0, false,
                                       j.stream_cap + j.nseg * pd::stuff_tile_bytes(), first_tile, tiles, c.e_stuff_state.as<unsigned long long>(),
                                       /*state_is_zero=*/attempt == 0, c.e_out.as<uint8_t>(), c.e_out.cap,
                                       reinterpret_cast<unsigned long long *>(c.h_totals), stream, nullptr, 0, &j.seg, debug().spin_budget));
        c.code_state_zero_words = code_words;
        HIP_TRY(hipStreamSynchronize(stream));
        if (c.h_totals[3]) return scan_retry_multipass(c);
        const uint64_t all_tiles = c.h_totals[2];
        if (all_tiles > first_tile + tiles) { // the guess was short: the tiles behind it, same buffers
            if (attempt > 2) return fail(PIXO_ERR_COMPRESSION, "Compression error: packed stream longer than announced");
            first_tile += tiles;
            tiles = all_tiles - first_tile;
            continue;
        }
        j.scan_bytes = c.h_totals[1];
        if (j.scan_bytes > c.e_out.cap) { // (unusually many 0xFF bytes: grow and repeat the stuffing pass only)
            if (attempt > 2) return fail(PIXO_ERR_COMPRESSION, "Compression error: stuffed stream larger than announced");
            want_cap = static_cast<size_t>(j.scan_bytes);
            first_tile = 0;
            tiles = all_tiles;
            continue;
        }
        j.nbytes = 0;
        return PIXO_OK;
    }
}

int scan_stuff_fused(Context &c, ScanJob &j, hipStream_t stream, uint64_t band_bit_offset, uint32_t *head, int *tail_bits,
                     uint32_t *tail, bool chained, HostTarget *host)
{
    namespace pd = pixo_dev;
    uint32_t shift = 0;
    if (j.band) {
        const uint64_t want = (8 - (band_bit_offset & 7)) & 7;
        j.head_bits = static_cast<int>(j.total_bits < want ? j.total_bits : want);
        shift = static_cast<uint32_t>(j.head_bits);
    }
    if (j.segmented) return scan_stuff_segmented(c, j, stream);
    // entropy-coded data holds a 0xFF every ~256 bytes; start from a quarter of the worst-case stream and grow on demand
    size_t want_cap = chained ? std::max<size_t>(j.stream_cap / 4, 4096) : static_cast<size_t>(j.nbytes + j.nbytes / 64 + 4096);
    // tiles: the exact number when the stream's length is known, otherwise a guess (64 bytes per block; noise at q = 80
    // has 28) — surplus workgroups leave at once, missing ones are launched below
    uint64_t first_tile = 0, tiles = chained ? pd::stuff_tiles(std::min<uint64_t>(j.stream_cap, j.n * 64 + 4096)) : pd::stuff_tiles(j.nbytes);
    for (int attempt = 0;; ++attempt) {
        uint8_t *out = nullptr;
        size_t out_cap = 0;
        if (host) {
            if (host->grow) {
                const int rc = c.reserve_hfile(host->before + want_cap + host->after);
                if (rc) return rc;
                host->p = c.h_file + host->before;
                host->cap = c.hfile_cap - host->before - host->after;
            }
            out = host->p;
            out_cap = host->cap;
        } else {
            HIP_TRY(c.e_out.reserve(want_cap));
            out = c.e_out.as<uint8_t>();
            out_cap = c.e_out.cap;
        }
        HIP_TRY(pd::launch_stuff_fused(c.e_stream.as<uint32_t>(), c.e_code_state.as<unsigned long long>(), j.code_state_words,
                                       shift, j.band, j.stream_cap, first_tile, tiles, c.e_stuff_state.as<unsigned long long>(),
                                       /*state_is_zero=*/chained && attempt == 0, out, out_cap,
                                       reinterpret_cast<unsigned long long *>(c.h_totals), stream, nullptr, 0, nullptr, debug().spin_budget));
        c.code_state_zero_words = j.code_state_words;
        // (no read-back copies: both kernels store their totals into the pinned mailbox h_totals — [0] bits of the scan,
        // [1] stuffed bytes, [2] packed bytes — which the host reads after the synchronisation below)
        uint32_t edge[3] = {0, 0, 0}; // band: stream word 0 (head bits) and the two words around the tail bits
        if (j.band) {
            const uint64_t tail_at = static_cast<uint64_t>(j.head_bits) + 8 * ((j.total_bits - j.head_bits) / 8);
            HIP_TRY(hipMemcpyAsync(&edge[0], c.e_stream.p, 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(&edge[1], c.e_stream.as<uint32_t>() + (tail_at >> 5), 8, hipMemcpyDeviceToHost, stream));
        }
        HIP_TRY(hipStreamSynchronize(stream));
        if (c.h_totals[3]) return scan_retry_multipass(c);
        j.total_bits = c.h_totals[0];
        const uint64_t packed = j.band ? (j.total_bits - j.head_bits) / 8 : (j.total_bits + 7) / 8;
        if (pd::stuff_tiles(packed) > first_tile + tiles) { // the guess was short: the tiles behind it, same buffers
            if (attempt > 2) return fail(PIXO_ERR_COMPRESSION, "Compression error: packed stream longer than announced");
            first_tile += tiles;
            tiles = pd::stuff_tiles(packed) - first_tile;
            continue;
        }
        j.scan_bytes = c.h_totals[1];
        j.nbytes = c.h_totals[2];
        if (j.scan_bytes > out_cap) { // (first call with unusually many 0xFF bytes: grow and repeat the stuffing pass only)
            if (host && !host->grow) return PIXO_OK; // (the caller's storage is what it is: the caller reports the size needed)
            if (attempt > 2) return fail(PIXO_ERR_COMPRESSION, "Compression error: stuffed stream larger than announced");
            want_cap = static_cast<size_t>(j.scan_bytes);
            first_tile = 0;
            tiles = pd::stuff_tiles(packed);
            continue;
        }
        if (j.band) {
            const int t = static_cast<int>((j.total_bits - j.head_bits) % 8);
            const uint64_t tail_at = static_cast<uint64_t>(j.head_bits) + 8 * ((j.total_bits - j.head_bits) / 8);
            *head = j.head_bits ? (edge[0] >> (32 - j.head_bits)) : 0u;
            *tail_bits = t;
            const uint64_t two = (static_cast<uint64_t>(edge[1]) << 32) | edge[2]; // MSB-first bits of the two words
            *tail = t ? static_cast<uint32_t>((two >> (64 - (tail_at & 31) - t)) & ((1u << t) - 1u)) : 0u;
        }
        return PIXO_OK;
    }
}

// Pack, 0xFF census, stuffing (+ restart markers): afterwards c.e_out holds j.scan_bytes finished bytes
// (one read-back).  A band starting at bit `band_bit_offset` of the scan is packed so that its whole bytes
// begin at word 1 of the stream: its first (8 - offset % 8) % 8 bits end word 0, the bits left over after the
// last whole byte follow it; both are returned unstuffed in head / tail (value, right-aligned).
int scan_pack(Context &c, ScanJob &j, hipStream_t stream, uint64_t band_bit_offset, uint32_t *head, int *tail_bits, uint32_t *tail)
{
    namespace pd = pixo_dev;
    if (j.fused) return scan_stuff_fused(c, j, stream, band_bit_offset, head, tail_bits, tail);
    uint64_t stream_bits = j.total_bits;
    uint32_t word_off = 0;
    if (j.band) {
        const uint64_t want = (8 - (band_bit_offset & 7)) & 7;
        j.head_bits = static_cast<int>(j.total_bits < want ? j.total_bits : want);
        j.a.bit_base = 32 - static_cast<uint32_t>(j.head_bits);
        j.nbytes = (j.total_bits - j.head_bits) / 8;
        stream_bits = j.a.bit_base + j.total_bits;
        word_off = 1;
    }
    const size_t stream_bytes = j.band ? ((stream_bits + 31) / 32 + 2) * 4 : (j.nbytes / 4 + 2) * 4;
    HIP_TRY(c.e_stream.reserve(stream_bytes));
    HIP_TRY(hipMemsetAsync(c.e_stream.p, 0, stream_bytes, stream));
    if (j.n) HIP_TRY(pd::launch_scan_pack(j.a, c.e_off.as<uint64_t>(), j.total_bits, j.nseg ? &j.plan : nullptr, c.e_stream.as<uint32_t>(), stream));
    const uint32_t *body = c.e_stream.as<uint32_t>() + word_off;
    const size_t tiles = pd::stuff_tile_count(j.nbytes);
    HIP_TRY(c.e_tile_ff.reserve((tiles ? tiles : 1) * 4));
    HIP_TRY(c.e_tile_base.reserve((tiles ? tiles : 1) * 8));
    if (tiles) HIP_TRY(pd::launch_ff_tile_count(body, j.nbytes, c.e_tile_ff.as<uint32_t>(), stream));
    HIP_TRY(pd::launch_exclusive_scan(c.e_tile_ff.as<uint32_t>(), tiles, c.e_tile_base.as<uint64_t>(),
                                      c.e_tmp.as<uint64_t>() + j.tmp_blocks + j.tmp_segs, c.e_totals.as<uint64_t>() + 1, stream));
    HIP_TRY(hipMemcpyAsync(c.h_totals + 1, c.e_totals.as<uint64_t>() + 1, 8, hipMemcpyDeviceToHost, stream));
    uint32_t edge[2] = {0, 0}; // band: word 0 (head bits) and the word holding the tail bits
    if (j.band) {
        HIP_TRY(hipMemcpyAsync(&edge[0], c.e_stream.p, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(&edge[1], c.e_stream.as<uint32_t>() + 1 + j.nbytes / 4, 4, hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    j.scan_bytes = j.nbytes + c.h_totals[1];
    HIP_TRY(c.e_out.reserve(j.scan_bytes ? j.scan_bytes : 1));
    if (tiles) HIP_TRY(pd::launch_stuff(body, j.nbytes, c.e_tile_base.as<uint64_t>(), c.e_out.as<uint8_t>(), stream));
    if (j.nseg) HIP_TRY(pd::launch_restart_markers(j.a, c.e_off.as<uint64_t>(), j.plan, c.e_stream.as<uint32_t>(), c.e_tile_base.as<uint64_t>(),
                                                   c.e_out.as<uint8_t>(), stream));
    if (j.band) {
        const int t = static_cast<int>((j.total_bits - j.head_bits) % 8);
        *head = j.head_bits ? (edge[0] & ((1u << j.head_bits) - 1u)) : 0u;
        *tail_bits = t;
        // the tail bits are the top bits of stream byte nbytes (MSB-first bytes inside big-endian words)
        const uint32_t byte = (edge[1] >> (24 - 8 * static_cast<uint32_t>(j.nbytes % 4))) & 0xFFu;
        *tail = t ? (byte >> (8 - t)) : 0u;
    }
    return PIXO_OK;
}

} // namespace pixo_capi

// MEASUREMENT only (bench.py, tools/): the DEVICE work of one baseline file — pixels -> the finished scan in the context's
// device buffer — enqueued on the caller's stream and not waited for: K calls back to back between two events give the device
// time per file without the call's host side (waits, the file's way over PCIe).  The kernels are the product's: the fused
// pixel -> scan kernel where it serves the job (*form = 1), else coefficient kernel + scan_code + the stuffing kernel on a grid
// sized like the product's first guess.  Nothing is delivered.
extern "C" int pixo_hip_debug_scan_device_async_batch(const void *d_pixels, const pixo_jpeg_options *options, uint32_t batch, void *stream_, int *form)
{
    using namespace pixo_capi;
    namespace pd = pixo_dev;
    PIXO_REQUIRE(d_pixels);
    PIXO_REQUIRE(options);
    std::string msg;
    int rc = pixo_host::validate(*options, false, 0, msg);
    if (rc) return fail(rc, msg);
    if (options->progressive || options->optimize_huffman) return fail(PIXO_ERR_COMPRESSION, "Compression error: pixo_hip_debug_scan_device_async measures baseline scans with standard tables");
    if (batch == 0 || batch > 65535) return fail(PIXO_ERR_COMPRESSION, "Compression error: batch must be 1..65535");
    Context *c = nullptr;
    if ((rc = context_on_current_device(&c))) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const pixo_host::Geometry g = pixo_host::geometry(options->width, options->height, options->color_type, options->subsampling);
    const size_t coef_bytes = (g.y_blocks + 2 * g.c_blocks) * 128 * batch;
    if ((rc = c->reserve_coef(coef_bytes))) return rc;
    int16_t *dy = static_cast<int16_t *>(c->d_coef), *dcb = dy + g.y_blocks * 64 * batch, *dcr = dcb + g.c_blocks * 64 * batch;
    ScanJob j;
    if (batch > 1) { // (the gap a batch leaves between two scans: EOI + the next file's headers)
        std::vector<uint8_t> probe_head;
        pixo_host::file_headers(probe_head, *options, pixo_host::HuffSet::standard());
        j.seg_gap = static_cast<uint32_t>(probe_head.size() + 2);
    }
    if ((rc = scan_begin(*c, j, dy, dcb, dcr, *options, g, batch, nullptr))) return rc;
    const bool fused = pixels_code_usable(j, *options, g, batch);
    if (!fused && !j.fused) return fail(PIXO_ERR_COMPRESSION, "Compression error: not a single-pass scan");
    if (form) *form = fused ? 1 : 0;
    if (fused) return scan_from_pixels(*c, j, *options, g, stream, d_pixels, nullptr, /*wait=*/false, batch); // (ONE kernel: the stuffed scans in c.e_out)
    {
        const float *qt_all = nullptr;
        if ((rc = device_tables(c->device, &qt_all))) return rc;
        HIP_TRY(pd::launch_jpeg_coeffs(d_pixels, options->width, options->height, g.gray, g.s420, batch, dy, g.gray ? nullptr : dcb,
                                       g.gray ? nullptr : dcr, qt_all + (options->quality - 1) * pixo_host::kDeviceQtFloats, stream));
        if ((rc = scan_lengths(*c, j, *options, g, stream, nullptr, /*wait=*/false))) return rc;
    }
    if (j.segmented) { // (the product's first guess of the stuffing grid: scan_stuff_segmented)
        const size_t code_words = pd::fused_code_state_words_seg(j.seg.nsegs, j.seg.blocks);
        const uint64_t per_seg = pd::stuff_tiles(static_cast<uint64_t>(j.seg.blocks) * 64 + 4096);
        HIP_TRY(c->e_out.reserve(std::max<size_t>(j.stream_cap / 4, 4096)));
        HIP_TRY(pd::launch_stuff_fused(c->e_stream.as<uint32_t>(), c->e_code_state.as<unsigned long long>(), code_words, 0, false,
                                       j.stream_cap + j.nseg * pd::stuff_tile_bytes(), 0, per_seg * j.nseg, c->e_stuff_state.as<unsigned long long>(),
                                       /*state_is_zero=*/true, c->e_out.as<uint8_t>(), c->e_out.cap, reinterpret_cast<unsigned long long *>(c->h_totals), stream,
                                       nullptr, 0, &j.seg, debug().spin_budget));
        c->code_state_zero_words = code_words;
        return PIXO_OK;
    }
    const size_t want_cap = std::max<size_t>(j.stream_cap / 4, 4096);
    HIP_TRY(c->e_out.reserve(want_cap));
    const uint64_t tiles = pd::stuff_tiles(std::min<uint64_t>(j.stream_cap, j.n * 64 + 4096));
    HIP_TRY(pd::launch_stuff_fused(c->e_stream.as<uint32_t>(), c->e_code_state.as<unsigned long long>(), j.code_state_words, 0, false, j.stream_cap, 0,
                                   tiles, c->e_stuff_state.as<unsigned long long>(), /*state_is_zero=*/true, c->e_out.as<uint8_t>(), c->e_out.cap,
                                   reinterpret_cast<unsigned long long *>(c->h_totals), stream, nullptr, 0, nullptr, debug().spin_budget));
    c->code_state_zero_words = j.code_state_words;
    return PIXO_OK;
}
extern "C" int pixo_hip_debug_scan_device_async(const void *d_pixels, const pixo_jpeg_options *options, void *stream_, int *form)
{
    return pixo_hip_debug_scan_device_async_batch(d_pixels, options, 1, stream_, form);
}
