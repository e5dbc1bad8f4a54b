// progressive.cpp — preset 2 ("max"): Huffman statistics of a tuple, the trellis-quantised tuple, the seven progressive
// scans of simple_progressive_script coded on the device.
#include "capi_internal.hpp"
#include "jpeg_trellis.hpp"

namespace pixo_capi {

// The seven scans of simple_progressive_script (progressive.rs:98-110) coded by the kernels of
// jpeg_entropy.hip over the device tuple; `out` already holds the file headers.  All scans are ONE
// packed stream of byte-aligned segments (the virtual block order is scan by scan, storage order inside
// a scan), so lengths / prefix sum / pack / 0xFF stuffing run once; the host only splices the seven SOS
// headers between the stuffed segments.
// The file is assembled in the context's pinned buffer: `head` (the file headers), then per scan its SOS header and its
// stuffed segment — every segment copied from the device straight to its final place — then EOI.  (Round 1 copied the
// stuffed stream to the host in one piece and spliced it into a std::vector, which the caller copied once more: two
// extra passes over the file through freshly mapped pages, about half of the 2.3 ms of a 4096x4096 preset-2 file.)
static int device_progressive_scans_multipass(const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_host::Geometry &g,
                                              const pixo_host::HuffSet &h, Context &c, const std::vector<uint8_t> &head, const uint8_t **file,
                                              size_t *file_len, uint8_t *pinned_dest, size_t dest_cap)
{
    namespace pd = pixo_dev;
    Stopwatch sw;
    hipStream_t stream = c.stream;
    pd::ProgArgs a;
    a.y = dy; a.cb = g.gray ? dy : dcb; a.cr = g.gray ? dy : dcr;
    const uint64_t size[7] = {g.y_blocks, g.c_blocks, g.c_blocks, g.y_blocks, g.y_blocks, g.c_blocks, g.c_blocks};
    a.first[0] = 0;
    for (int i = 0; i < 7; ++i) a.first[i + 1] = a.first[i] + size[i];
    const uint64_t n = a.first[7];
    HIP_TRY(c.e_tables.reserve(pixo_scan::kScanTableUpload * 4));
    HIP_TRY(c.g_flags.reserve(n * 4));
    HIP_TRY(c.g_rank.reserve(n * 8));
    HIP_TRY(c.g_by_rank.reserve(n * 4));
    HIP_TRY(c.e_len.reserve(n * 4));
    HIP_TRY(c.e_off.reserve(n * 8));
    HIP_TRY(c.e_seg_bytes.reserve(8 * 8));
    HIP_TRY(c.e_seg_off.reserve(8 * 8));
    // a block of an AC scan: at most 63 * 26 bits + an end-of-band run of at most 16 + 14 bits
    const size_t tmp_blocks = pd::scan_tile_count(n) + 1, tmp_segs = pd::scan_tile_count(7) + 1;
    const size_t tmp_tiles = pd::scan_tile_count(pd::stuff_tile_count(n * 212 + 64)) + 1;
    HIP_TRY(c.e_tmp.reserve((tmp_blocks + tmp_segs + tmp_tiles) * 8));
    HIP_TRY(c.e_totals.reserve(32));
    { const int rc_t = c.ensure_totals(); if (rc_t) return rc_t; }
    a.tables = c.e_tables.as<uint32_t>();
    a.flags = c.g_flags.as<uint32_t>();
    a.nonempty = c.e_len.as<uint32_t>(); // only the input of the rank prefix sum: the lengths reuse it
    a.rank = c.g_rank.as<uint64_t>();
    a.by_rank = c.g_by_rank.as<uint32_t>();

    uint32_t packed[pixo_host::kScanTableWords];
    pixo_host::pack_scan_tables(h, packed);
    for (uint32_t &w : packed) // progressive.rs:363-381: a symbol the table lacks is coded as (0, 4 bits)
        if ((w >> 16) == 0) w = 4u << 16;
    { const int rc = upload_scan_tables(c, packed, stream); if (rc) return rc; }
    uint64_t *totals = c.e_totals.as<uint64_t>();
    HIP_TRY(pd::launch_prog_flags(a, stream));
    HIP_TRY(pd::launch_exclusive_scan(a.nonempty, n, c.g_rank.as<uint64_t>(), c.e_tmp.as<uint64_t>(), totals + 2, stream));
    HIP_TRY(pd::launch_prog_by_rank(a, stream));
    HIP_TRY(pd::launch_prog_lengths(a, c.e_len.as<uint32_t>(), stream));
    HIP_TRY(pd::launch_exclusive_scan(c.e_len.as<uint32_t>(), n, c.e_off.as<uint64_t>(), c.e_tmp.as<uint64_t>(), totals, stream));
    HIP_TRY(pd::launch_prog_segment_sizes(a, c.e_off.as<uint64_t>(), totals, c.e_seg_bytes.as<uint32_t>(), stream));
    HIP_TRY(pd::launch_exclusive_scan(c.e_seg_bytes.as<uint32_t>(), 7, c.e_seg_off.as<uint64_t>(), c.e_tmp.as<uint64_t>() + tmp_blocks,
                                      totals + 1, stream));
    HIP_TRY(hipMemcpyAsync(c.h_totals, totals, 16, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    sw.lap("prog flags+rank+lengths");
    const uint64_t total_bits = c.h_totals[0], nbytes = c.h_totals[1];
    const size_t stream_bytes = (nbytes / 4 + 2) * 4;
    HIP_TRY(c.e_stream.reserve(stream_bytes));
    HIP_TRY(hipMemsetAsync(c.e_stream.p, 0, stream_bytes, stream));
    HIP_TRY(pd::launch_prog_pack(a, c.e_off.as<uint64_t>(), total_bits, c.e_seg_off.as<uint64_t>(), c.e_stream.as<uint32_t>(), stream));
    const size_t tiles = pd::stuff_tile_count(nbytes);
    HIP_TRY(c.e_tile_ff.reserve(tiles * 4));
    HIP_TRY(c.e_tile_base.reserve(tiles * 8));
    HIP_TRY(pd::launch_ff_tile_count(c.e_stream.as<uint32_t>(), nbytes, c.e_tile_ff.as<uint32_t>(), stream));
    HIP_TRY(pd::launch_exclusive_scan(c.e_tile_ff.as<uint32_t>(), tiles, c.e_tile_base.as<uint64_t>(),
                                      c.e_tmp.as<uint64_t>() + tmp_blocks + tmp_segs, totals + 1, stream));
    HIP_TRY(hipMemcpyAsync(c.h_totals + 1, totals + 1, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    sw.lap("prog pack+ff census");
    const uint64_t scan_bytes = nbytes + c.h_totals[1];
    HIP_TRY(c.e_out.reserve(scan_bytes + 16));
    HIP_TRY(pd::launch_stuff(c.e_stream.as<uint32_t>(), nbytes, c.e_tile_base.as<uint64_t>(), c.e_out.as<uint8_t>(), stream));
    const pd::SegmentPlan plan{7, c.e_seg_off.as<uint64_t>()};
    HIP_TRY(pd::launch_segment_out_offsets(plan, nbytes, c.e_stream.as<uint32_t>(), c.e_tile_base.as<uint64_t>(),
                                           c.g_rank.as<uint64_t>(), stream)); // the rank array is free again
    uint64_t start[8];
    HIP_TRY(hipMemcpyAsync(start, c.g_rank.p, 7 * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    start[7] = scan_bytes;
    for (int i = 6; i >= 0; --i)
        if (start[i] == ~0ull) start[i] = start[i + 1]; // empty scans at the end of the stream
    const size_t total = head.size() + 7 * 10 + scan_bytes + 2;
    uint8_t *p = pinned_dest; // the caller's pinned storage when the file fits: the seven copies below are its only pass over the bytes
    if (!p || total > dest_cap) {
        int rc = c.reserve_hfile(total);
        if (rc) return rc;
        p = c.h_file;
    }
    std::memcpy(p, head.data(), head.size());
    size_t pos = head.size();
    static const uint8_t script[7][3] = {{0, 0, 0}, {1, 0, 0}, {2, 0, 0}, {0, 1, 10}, {0, 11, 63}, {1, 1, 63}, {2, 1, 63}};
    for (int i = 0; i < 7; ++i) { // write_sos_progressive, jpeg/mod.rs:650-682
        const uint8_t sos[10] = {0xFF, 0xDA, 0, 8, 1, static_cast<uint8_t>(script[i][0] + 1),
                                 static_cast<uint8_t>(script[i][0] == 0 ? 0x00 : 0x11), script[i][1], script[i][2], 0};
        std::memcpy(p + pos, sos, 10);
        pos += 10;
        const size_t n = static_cast<size_t>(start[i + 1] - start[i]);
        if (n) HIP_TRY(hipMemcpyAsync(p + pos, c.e_out.as<uint8_t>() + start[i], n, hipMemcpyDeviceToHost, stream));
        pos += n;
    }
    p[pos] = 0xFF; p[pos + 1] = 0xD9;
    HIP_TRY(hipStreamSynchronize(stream));
    *file = p;
    *file_len = pos + 2;
    sw.lap("prog stuff+copy+splice");
    return PIXO_OK;
}

// write_sos_progressive (jpeg/mod.rs:650-682) of scan i of simple_progressive_script (progressive.rs:98-110)
static void sos_of_scan(int i, uint8_t sos[10])
{
    static const uint8_t script[7][3] = {{0, 0, 0}, {1, 0, 0}, {2, 0, 0}, {0, 1, 10}, {0, 11, 63}, {1, 1, 63}, {2, 1, 63}};
    const uint8_t v[10] = {0xFF, 0xDA, 0, 8, 1, static_cast<uint8_t>(script[i][0] + 1), static_cast<uint8_t>(script[i][0] == 0 ? 0x00 : 0x11),
                           script[i][1], script[i][2], 0};
    std::memcpy(sos, v, 10);
}

// The same scans in ONE pass over the tuple (round 4): prog_code_kernel (every scan a byte-aligned segment with a packed stream
// of its own), the segment layout, and stuff_fused<SEG>, which leaves the seven stuffed scans in c.e_out at their FINAL
// spacing — ten free bytes between two of them, where the next scan's SOS header goes — so that the file's entropy-coded
// part crosses PCIe as one copy (a gray image, whose chroma scans are empty and yet have headers: one copy per run of scans
// that lie at their final spacing) and the host only fills in the headers.  Three launches and one synchronisation, where
// the multi-pass form above has about twenty launches, three synchronisations and one copy per scan.
// kRetryMultipass: a look-back gave up waiting (the caller runs the multi-pass form).
static int device_progressive_scans_fused(const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_host::Geometry &g,
                                          const pixo_host::HuffSet &h, Context &c, const std::vector<uint8_t> &head, const uint8_t **file,
                                          size_t *file_len, uint8_t *pinned_dest, size_t dest_cap)
{
    namespace pd = pixo_dev;
    Stopwatch sw;
    hipStream_t stream = c.stream;
    const uint64_t size[7] = {g.y_blocks, g.c_blocks, g.c_blocks, g.y_blocks, g.y_blocks, g.c_blocks, g.c_blocks};
    pd::ProgCode a{};
    pd::SegArgs sg;
    a.y = dy; a.cb = g.gray ? dy : dcb; a.cr = g.gray ? dy : dcr;
    size_t stream_bytes = 0;
    uint64_t blocks_all = 0;
    a.first_group[0] = 0;
    for (uint32_t i = 0; i < 7; ++i) {
        if (!size[i]) continue; // (the chroma scans of a gray image: headers only)
        const uint32_t k = a.nscans++;
        a.scan_id[k] = i;
        a.size[k] = size[i];
        a.first_group[k + 1] = a.first_group[k] + pd::prog_groups(i, size[i]);
        sg.var_word[k] = stream_bytes / 4;
        stream_bytes += pd::prog_stream_bytes(i, size[i]);
        blocks_all += size[i];
    }
    const uint64_t groups = a.first_group[a.nscans];
    pd::prog_code_plan(a);
    sg.nsegs = a.nscans;
    sg.var = 1;
    sg.marker_bytes = 10; // room for the next scan's SOS header
    HIP_TRY(c.e_tables.reserve(pixo_scan::kScanTableUpload * 4));
    HIP_TRY(c.e_stream.reserve(stream_bytes + 64));
    const size_t state_words = pd::prog_code_state_words(groups);
    if (state_words * 8 > c.e_code_state.cap) c.code_state_zero_words = 0; // (a new buffer)
    HIP_TRY(c.e_code_state.reserve(state_words * 8));
    const size_t stuff_words = pd::fused_stuff_state_words(stream_bytes) + 8;
    HIP_TRY(c.e_stuff_state.reserve(stuff_words * 8));
    HIP_TRY(c.e_segs.reserve((4 * 8 + 2) * 8));
    unsigned long long *base = c.e_segs.as<unsigned long long>();
    sg.bits = base; sg.layout = base + 8; sg.bytes = base + 2 * 8 + 2; sg.out_end = base + 3 * 8 + 2;
    { const int rc_s = c.reserve_hsegs(pixo_host::kScanTableWords); if (rc_s) return rc_s; } // (8 scan ends; and the counters' room: see scan_begin)
    sg.host_out_end = reinterpret_cast<unsigned long long *>(c.h_segs);
    { const int rc_t = c.ensure_totals(); if (rc_t) return rc_t; }
    uint32_t packed[pixo_host::kScanTableWords];
    pixo_host::pack_scan_tables(h, packed);
    for (uint32_t &w : packed) // progressive.rs:363-381: a symbol the table lacks is coded as (0, 4 bits)
        if ((w >> 16) == 0) w = 4u << 16;
    { const int rc = upload_scan_tables(c, packed, stream); if (rc) return rc; }
    a.tables = c.e_tables.as<uint32_t>();
    const bool zero = c.code_state_zero_words >= state_words;
    c.code_state_zero_words = 0;
    unsigned long long *mailbox = reinterpret_cast<unsigned long long *>(c.h_totals);
    HIP_TRY(pd::launch_prog_code(a, sg, c.e_code_state.as<unsigned long long>(), zero, c.e_stream.as<uint32_t>(),
                                 c.e_stuff_state.as<unsigned long long>(), stuff_words, mailbox, stream, debug().spin_budget));
    // (the scans' layout — bytes and 16 KiB tiles of each — is worked out by the stuffing kernel's workgroups themselves: seg.var)
    // tiles: a guess of 40 bytes per (scan, block) pair + one partial tile per scan; the stuffing kernel says how many there are
    uint64_t first_tile = 0, tiles = pd::stuff_tiles(blocks_all * 40 + 4096) + a.nscans;
    size_t want_cap = std::max<size_t>(stream_bytes / 4, 4096);
    uint64_t scan_bytes = 0;
    // Small files (round 5; like the baseline path, pieces.cpp): the stuffing kernel stores the scans straight into host memory the
    // GPU can write — the caller's pinned storage or the context's pinned file buffer — at their places in the FILE; no copy and no
    // second wait.  "Small": the last progressive file of this context was (768 KB); all seven scans present (the gaps between
    // them are then exactly the SOS headers' places).  Too small a destination: the stuffing pass runs again into device memory.
    constexpr uint64_t kDirectBytes = 768u << 10;
    const size_t body_at = head.size() + 10; // the first scan's first byte in the file
    uint8_t *direct_host = nullptr, *direct_dev = nullptr;
    size_t direct_cap = 0;
    if (a.nscans == 7 && !debug().no_direct_small && c.last_prog_bytes && c.last_prog_bytes <= kDirectBytes) {
        const size_t guess = body_at + 2 * static_cast<size_t>(c.last_prog_bytes) + 4096;
        if (pinned_dest && dest_cap >= guess) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, pinned_dest) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) {
                direct_host = pinned_dest; direct_dev = static_cast<uint8_t *>(at.devicePointer); direct_cap = dest_cap;
            } else (void)hipGetLastError();
        }
        if (!direct_host) {
            const int rc_f = c.reserve_hfile(guess);
            if (rc_f) return rc_f;
            direct_host = direct_dev = c.h_file; direct_cap = c.hfile_cap;
        }
    }
    for (int attempt = 0;; ++attempt) {
        uint8_t *out = nullptr;
        size_t out_cap = 0;
        if (direct_host) { out = direct_dev + body_at; out_cap = direct_cap - body_at - 2; }
        else { HIP_TRY(c.e_out.reserve(want_cap)); out = c.e_out.as<uint8_t>(); out_cap = c.e_out.cap; }
        HIP_TRY(pd::launch_stuff_fused(c.e_stream.as<uint32_t>(), c.e_code_state.as<unsigned long long>(), state_words, 0, false,
                                       stream_bytes + 8 * pd::stuff_tile_bytes(), first_tile, tiles, c.e_stuff_state.as<unsigned long long>(),
                                       /*state_is_zero=*/attempt == 0, out, out_cap, mailbox, stream, nullptr, 0, &sg,
                                       debug().spin_budget));
        c.code_state_zero_words = state_words;
        HIP_TRY(hipStreamSynchronize(stream));
        if (c.h_totals[3]) return scan_retry_multipass(c);
        const uint64_t all_tiles = c.h_totals[2];
        if (all_tiles > first_tile + tiles) { // the guess was short: the tiles behind it, same buffers
            if (attempt > 2) return fail(PIXO_ERR_COMPRESSION, "Compression error: packed stream longer than announced");
            first_tile += tiles;
            tiles = all_tiles - first_tile;
            continue;
        }
        scan_bytes = c.h_totals[1];
        if (scan_bytes > out_cap) { // (unusually large: grow and repeat the stuffing pass only — into device memory)
            if (attempt > 3) return fail(PIXO_ERR_COMPRESSION, "Compression error: stuffed stream larger than announced");
            direct_host = nullptr;
            want_cap = static_cast<size_t>(scan_bytes);
            first_tile = 0;
            tiles = all_tiles;
            continue;
        }
        break;
    }
    c.last_prog_bytes = scan_bytes;
    sw.lap("prog code+stuff");
    // c.e_out (or the file itself): [scan k0 | 10 | scan k1 | 10 | ...]; c.h_segs[k]: where scan k's bytes end.  The file: head, then
    // per scan of the script its SOS header and (if it has blocks) its bytes, then EOI.
    uint64_t dev_begin[7], dev_end[7];
    for (uint32_t k = 0; k < a.nscans; ++k) {
        dev_end[k] = c.h_segs[k];
        dev_begin[k] = k ? c.h_segs[k - 1] + sg.marker_bytes : 0;
    }
    const size_t payload = static_cast<size_t>(scan_bytes - static_cast<uint64_t>(sg.marker_bytes) * (a.nscans - 1));
    const size_t total = head.size() + 7 * 10 + payload + 2;
    if (direct_host) { // the scans already lie in the file: headers into the gaps, EOI behind
        uint8_t *p = direct_host;
        std::memcpy(p, head.data(), head.size());
        sos_of_scan(0, p + head.size());
        for (uint32_t k = 1; k < 7; ++k) sos_of_scan(static_cast<int>(k), p + body_at + dev_end[k - 1]);
        p[total - 2] = 0xFF; p[total - 1] = 0xD9;
        *file = p;
        *file_len = total;
        sw.lap("prog headers");
        return PIXO_OK;
    }
    uint8_t *p = pinned_dest;
    if (!p || total > dest_cap) {
        const int rc = c.reserve_hfile(total);
        if (rc) return rc;
        p = c.h_file;
    }
    std::memcpy(p, head.data(), head.size());
    size_t pos = head.size();
    size_t file_at[7]; // where scan k's bytes go
    size_t sos_at[7];
    { uint32_t k = 0;
      for (int i = 0; i < 7; ++i) {
          sos_at[i] = pos; pos += 10;
          if (size[i]) { file_at[k] = pos; pos += static_cast<size_t>(dev_end[k] - dev_begin[k]); ++k; }
      } }
    // one copy per run of scans whose spacing in c.e_out is their spacing in the file
    for (uint32_t k = 0; k < a.nscans;) {
        uint32_t e = k;
        while (e + 1 < a.nscans && file_at[e + 1] - file_at[k] == dev_begin[e + 1] - dev_begin[k]) ++e;
        const size_t n = static_cast<size_t>(dev_end[e] - dev_begin[k]);
        if (n) HIP_TRY(hipMemcpyAsync(p + file_at[k], c.e_out.as<uint8_t>() + dev_begin[k], n, hipMemcpyDeviceToHost, stream));
        k = e + 1;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    for (int i = 0; i < 7; ++i) sos_of_scan(i, p + sos_at[i]); // (after the copies: a run's copy passes over the gaps)
    p[pos] = 0xFF; p[pos + 1] = 0xD9;
    *file = p;
    *file_len = pos + 2;
    sw.lap("prog copy+headers");
    return PIXO_OK;
}

int device_progressive_scans(const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_host::Geometry &g,
                             const pixo_host::HuffSet &h, Context &c, const std::vector<uint8_t> &head, const uint8_t **file,
                             size_t *file_len, uint8_t *pinned_dest, size_t dest_cap)
{
    if (!debug().multipass_entropy) {
        const int rc = device_progressive_scans_fused(dy, dcb, dcr, g, h, c, head, file, file_len, pinned_dest, dest_cap);
        if (rc != kRetryMultipass) return rc;
    }
    return device_progressive_scans_multipass(dy, dcb, dcr, g, h, c, head, file, file_len, pinned_dest, dest_cap);
}

// Huffman tables of a file over the device tuple: the standard ones, or (optimize_huffman) those built
// from the statistics of a baseline walk (build_optimized_huffman_tables, jpeg/mod.rs:684-824) — counted
// on the device, constructed on the host.
int huffman_for_tuple(const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
                      const pixo_host::Geometry &g, Context &c, pixo_host::HuffSet &h)
{
    namespace pd = pixo_dev;
    h = pixo_host::HuffSet::standard();
    if (!o.optimize_huffman) return PIXO_OK;
    pd::ScanArgs a;
    a.y = dy; a.cb = dcb; a.cr = dcr; a.tables = nullptr;
    a.mode = g.gray ? 0 : (g.s420 ? 2 : 1);
    a.nblocks = g.y_blocks + 2 * g.c_blocks;
    a.blocks_per_mcu = g.gray ? 1 : (g.s420 ? 6 : 3);
    a.marker_bytes = 2;
    a.restart = scan_has_restart_markers(o, g) ? o.restart_interval : 0;
    a.seed_dc[0] = a.seed_dc[1] = a.seed_dc[2] = 0; a.bit_base = 0; a.pad_last = 1;
    HIP_TRY(c.e_hist.reserve(pixo_host::kScanTableWords * 8));
    HIP_TRY(c.e_count.reserve(pd::scan_count_scratch_bytes()));
    HIP_TRY(pd::launch_scan_count(a, c.e_count.as<uint32_t>(), c.e_hist.as<unsigned long long>(), c.stream));
    uint64_t counts[pixo_host::kScanTableWords];
    HIP_TRY(hipMemcpyAsync(counts, c.e_hist.p, sizeof counts, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    uint64_t dc[2][12], ac[2][256];
    split_counts(counts, dc, ac);
    h = pixo_host::HuffSet::optimized(dc, ac, !g.gray);
    return PIXO_OK;
}

// Progressive files (jpeg/mod.rs:397-419, :872-927).  Device pixels -> file in `out`:
//   tables   optimised ones come from the statistics of a BASELINE walk over the PLAIN quantiser's
//            coefficients (build_optimized_huffman_tables, :684-824, never uses trellis): the ordinary
//            coefficient kernel + the device histogram pass;
//   tuple    `trellis_quant`: the coefficient kernel in raw mode (unquantised transform) followed by the
//            trellis kernel; otherwise the ordinary kernel (`trellis_quant` acts nowhere else: a baseline
//            encode with the flag set is an ordinary baseline encode, encode_scan never reads it);
//   scans    device_progressive_scans above (PIXO_HIP_HOST_ENTROPY=1: the host twin in jpeg_host.cpp on a
//            pinned copy of the tuple).
// Progressive file from device pixels; *file points into the context's pinned buffer (or into `spill`: the host twin).
int progressive_to_view(const void *d_pixels, const pixo_jpeg_options &o, const pixo_host::Geometry &g, Context &c,
                        std::vector<uint8_t> &spill, const uint8_t **file, size_t *file_len, uint8_t *pinned_dest, size_t dest_cap)
{
    namespace pd = pixo_dev;
    int rc;
    int16_t *dy = nullptr, *dcb = nullptr, *dcr = nullptr;
    const float *qt_all = nullptr;
    if ((rc = device_tables(c.device, &qt_all))) return rc;
    const float *qt = qt_all + (o.quality - 1) * pixo_host::kDeviceQtFloats;
    pixo_host::HuffSet h;
    const bool need_plain = o.optimize_huffman || !o.trellis_quant;
    const bool late_tables = o.optimize_huffman && o.trellis_quant && !debug().host_entropy;
    // Round 5, small images (the trellis kernel's wavefronts — 64 blocks each — do not fill the chip's 1,024 SIMDs: up to 1080p):
    // the statistics run on a SECOND stream beside the raw transform + the search, whose 117 us are the latency of one
    // wavefront's 63 steps whatever the image's size.  (On large images the search keeps every SIMD issuing and kernels beside
    // it only take its slots: measured in round 4, not kept.)  The plain tuple then has a buffer of its own — the search
    // writes the file's tuple while the statistics still read theirs.
    constexpr size_t kSideStatsBlocks = 65536;
    const bool side_stats = late_tables && !debug().no_side_stats && g.y_blocks + 2 * g.c_blocks <= kSideStatsBlocks;
    hipStream_t stats_stream = c.stream;
    if (side_stats) {
        if (!c.copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking));
        if (!c.side_ready) HIP_TRY(hipEventCreateWithFlags(&c.side_ready, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c.side_ready, c.stream)); // (the context's stream is ordered behind the pixels' producer)
        HIP_TRY(hipStreamWaitEvent(c.copy_stream, c.side_ready, 0));
        stats_stream = c.copy_stream;
        HIP_TRY(c.t_plain.reserve((g.y_blocks + 2 * g.c_blocks) * 128));
        dy = c.t_plain.as<int16_t>(); dcb = dy + g.y_blocks * 64; dcr = dcb + g.c_blocks * 64;
        if ((rc = coeffs_rows(c, d_pixels, o, g, stats_stream, dy, g.gray ? nullptr : dcb, g.gray ? nullptr : dcr, 0, 0))) return rc;
    } else if (need_plain && (rc = coeffs_on_device(c, d_pixels, o, g, c.stream, &dy, &dcb, &dcr))) return rc;
    // Preset 2 (optimised tables AND trellis): the statistics (plain tuple -> baseline walk -> counts) are enqueued, the counts
    // start their way to a pinned buffer, and the raw transform + trellis search follow on the same stream AT ONCE: the host
    // waits for the counts' event only and builds the tables while the search (0.29 ms for 4096x4096) runs.  Rounds 1-3
    // synchronised, built the tables and only then launched the search: 30 us of idle GPU per file.
    if (late_tables) {
        pd::ScanArgs a;
        a.y = dy; a.cb = dcb; a.cr = dcr; a.tables = nullptr;
        a.mode = g.gray ? 0 : (g.s420 ? 2 : 1);
        a.nblocks = g.y_blocks + 2 * g.c_blocks;
        a.blocks_per_mcu = g.gray ? 1 : (g.s420 ? 6 : 3);
        a.marker_bytes = 2;
        a.restart = scan_has_restart_markers(o, g) ? o.restart_interval : 0;
        a.seed_dc[0] = a.seed_dc[1] = a.seed_dc[2] = 0; a.bit_base = 0; a.pad_last = 1;
        HIP_TRY(c.e_hist.reserve(pixo_host::kScanTableWords * 8));
        HIP_TRY(c.e_count.reserve(pd::scan_count_scratch_bytes()));
        HIP_TRY(pd::launch_scan_count(a, c.e_count.as<uint32_t>(), c.e_hist.as<unsigned long long>(), stats_stream));
        if ((rc = c.reserve_hsegs(pixo_host::kScanTableWords))) return rc;
        HIP_TRY(hipMemcpyAsync(c.h_segs, c.e_hist.p, pixo_host::kScanTableWords * 8, hipMemcpyDeviceToHost, stats_stream));
        if (!c.stats_done) HIP_TRY(hipEventCreateWithFlags(&c.stats_done, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c.stats_done, stats_stream));
    } else if ((rc = huffman_for_tuple(dy, dcb, dcr, o, g, c, h))) return rc;
    const size_t blocks = g.y_blocks + 2 * g.c_blocks, coef_bytes = blocks * 128;
    if (o.trellis_quant) {
        HIP_TRY(c.t_raw.reserve((blocks + 63) / 64 * 64 * 256)); // (whole wavefronts of the trellis kernel: jpeg_kernels.hpp)
        if ((rc = c.reserve_coef(coef_bytes))) return rc;
        float *ry = c.t_raw.as<float>(), *rcb = ry + g.y_blocks * 64, *rcr = rcb + g.c_blocks * 64;
        dy = static_cast<int16_t *>(c.d_coef); dcb = dy + g.y_blocks * 64; dcr = dcb + g.c_blocks * 64;
        HIP_TRY(pd::launch_jpeg_coeffs(d_pixels, o.width, o.height, g.gray, g.s420, 1, ry, g.gray ? nullptr : rcb,
                                       g.gray ? nullptr : rcr, qt, c.stream, /*raw_f32=*/true));
        // one launch over the whole tuple (the planes are contiguous): luminance steps, then chrominance steps
        HIP_TRY(c.t_trail.reserve(pd::trellis_scratch_bytes(blocks)));
        HIP_TRY(pd::launch_trellis(ry, qt + 128, qt + 192, dy, blocks, g.y_blocks, c.t_trail.p, c.stream));
    }
    if (late_tables) { // the counts have arrived (the search is still running): tables, exactly like optimized_from_counts
        HIP_TRY(hipEventSynchronize(c.stats_done));
        uint64_t dc[2][12], ac[2][256];
        split_counts(c.h_segs, dc, ac);
        h = pixo_host::HuffSet::optimized(dc, ac, !g.gray);
    }
    if (!debug().host_entropy) {
        std::vector<uint8_t> head;
        pixo_host::file_headers(head, o, h);
        return device_progressive_scans(dy, dcb, dcr, g, h, c, head, file, file_len, pinned_dest, dest_cap);
    }
    if ((rc = c.reserve_hcoef(coef_bytes))) return rc;
    HIP_TRY(hipMemcpyAsync(c.h_coef, dy, coef_bytes, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    const int16_t *hy = static_cast<const int16_t *>(c.h_coef), *hcb = hy + g.y_blocks * 64, *hcr = hcb + g.c_blocks * 64;
    pixo_host::encode_progressive_file(hy, hcb, hcr, o, h, spill);
    *file = spill.data();
    *file_len = spill.size();
    return PIXO_OK;
}

} // namespace pixo_capi
