// pieces.cpp — whole baseline files from a device tuple (or from pixels not transformed yet): a scan coded in PIECES so
// that the file's way to the host overlaps the coding, the one-piece path, delivery into pinned / caller / malloc'd memory.
#include <malloc.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>

#include "capi_internal.hpp"

namespace pixo_capi {

// One uninterrupted scan of a large image, coded in PIECES so that the file's way to the host (0.21 of the 0.31 ms of a
// 4096x4096 noise image, 3.3 of 4.4 ms for 16384x16384) overlaps the coding: piece k = a run of consecutive groups,
// coded and stuffed by its own launch pair; the pieces hand each other the bit position and the byte position on the
// device (pixo_dev::ScanPiece), the host only waits for piece k's event to learn how many bytes it may copy — on a second
// stream — while piece k + 1 is being coded.  The bytes are the same as from one launch pair; a piece whose stream
// outgrows the guess its stuffing grid was sized for, or an output buffer that proves too small, sends the caller back
// to the one-piece path (return value 1; nothing of the result is kept).
constexpr uint32_t kMaxPieces = 16;
// Piece sizes (debug switches piece_groups / piece_medium / piece_schedule, capi_internal.hpp):
//  * a piece is at least debug().piece_groups (2048) groups of 192 blocks, and a scan of fewer than two such pieces is not cut
//    into equal pieces.  2048 groups = the scan of a 4096x4096 4:2:0 image: the kernels of a smaller piece are mostly
//    start-up — a sixth of that scan takes 28 us where the whole takes 52 — and a 4096x4096 image in 2 or 6 equal pieces is
//    no faster than in one (0.30-0.33 against 0.31 ms); a 16384x16384 scan in 16 such pieces hides its 1 ms of coding
//    behind 3.4 ms of PCIe;
//  * medium scans (debug().piece_medium (1024) <= groups < 2 piece_groups): a few pieces that GROW, relative sizes
//    debug().piece_schedule ("1:3").  4096x4096 noise, 11 MB file, into pinned memory: one piece 0.313 ms, "1:3" 0.301,
//    "1:2:5" 0.302, "1:2:3:4" 0.315; with the coefficient kernel band by band as well: "1:3" 0.291, "1:2:5" 0.299, "1:5" 0.310
//    (tools/gpu/r2w.sh): the first piece's bytes leave 60 us after the start instead of 100, the rest is the file's
//    0.21 ms on PCIe.
inline bool pieces_enabled() { return !debug().one_piece; }
inline uint64_t piece_min_groups() { return debug().piece_groups; }
inline const std::vector<uint32_t> &piece_schedule() { return debug().piece_schedule; }
inline bool piece_medium_forced() { return debug().piece_medium_forced; }
inline uint64_t piece_medium_groups() { return debug().piece_medium; }
inline bool direct_host_stores() { return debug().direct_stores; }

// Pixels whose coefficients have not been computed yet (the tuple's place is reserved, j.a points at it): the entropy
// stage launches the coefficient kernel itself — for a scan coded in pieces, band by band in front of each piece, so
// that the first piece's bytes can leave before the rest of the image has even been transformed.

} // namespace pixo_capi

// A context's second host thread.  A copy from or to PAGEABLE host memory keeps the calling thread inside the runtime until
// the bytes have moved (it stages them through pinned buffers): one thread can therefore not keep both directions of the
// PCIe link busy — 48 MiB in + 11 MiB out take 1.11 ms from one thread and 0.93 ms from two (tools/ubench/duplex.cpp,
// profiles/r03_duplex_copies.txt).  The helper runs one job at a time for its context; it lives as long as the process.
struct pixo_capi::CopyHelper {
    std::mutex m;
    std::condition_variable cv, idle;
    std::function<void()> job;
    bool has_job = false, busy = false, stop = false;
    std::thread th;
    CopyHelper() : th([this] { loop(); }) {}
    // (a context that is deleted — excess contexts of the pool, pixo_hip_trim — takes its helper thread with it;
    // it used to stay behind, blocked on its condition variable, one per deleted context)
    ~CopyHelper()
    {
        {
            std::lock_guard<std::mutex> lock(m);
            stop = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
    void loop()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] { return has_job || stop; });
                if (!has_job) return; // (stop)
                f.swap(job);
                has_job = false;
            }
            f();
            {
                std::lock_guard<std::mutex> lock(m);
                busy = false;
            }
            idle.notify_all();
        }
    }
    void start(std::function<void()> f)
    {
        {
            std::lock_guard<std::mutex> lock(m);
            job = std::move(f);
            has_job = busy = true;
        }
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lock(m);
        idle.wait(lock, [&] { return !busy; });
    }
};

namespace pixo_capi {

void destroy_copy_helper(CopyHelper *h) { delete h; }

int device_entropy_pieces(Context &c, ScanJob &j, hipStream_t stream, uint8_t *dst, size_t dst_cap, uint64_t *scan_bytes,
                          const PixelSource *src)
{ // dst: where the stuffed scan goes on the host (dst_cap bytes available); tables are uploaded, j.a is set up
    namespace pd = pixo_dev;
    const uint64_t kGroupBlocks = 192, groups = (j.n + kGroupBlocks - 1) / kGroupBlocks;
    // where the pieces begin (in groups).  A large scan: equal pieces of at least piece_min_groups().  A medium one (a
    // 4096x4096 image): a few pieces that GROW — the first small, so that its bytes leave early, each next one coded
    // while the one before travels (weights from piece_schedule()).
    uint64_t begin[kMaxPieces + 1];
    uint32_t pieces = 0;
    const bool from_host = src && src->host_px;
    if (from_host) { // pixels still in host memory: pieces sized for the UPLOAD pipeline — about 12 MB of pixels each, so that
        // a band's kernels (tens of microseconds) disappear behind the next band's way over PCIe (>= 100 us)
        const uint64_t px_bytes = static_cast<uint64_t>(src->o->width) * src->o->height * (src->g->gray ? 1 : 3);
        const uint32_t want = static_cast<uint32_t>(std::min<uint64_t>(kMaxPieces, std::max<uint64_t>(2, px_bytes / (uint64_t{debug().bands_upload_mb} << 20))));
        for (uint32_t k = 0; k < want; ++k) {
            const uint64_t g0 = groups * k / want;
            if (pieces == 0 || g0 > begin[pieces - 1]) begin[pieces++] = g0;
        }
    } else if (groups >= 2 * piece_min_groups()) {
        const uint64_t per = std::max<uint64_t>((groups + kMaxPieces - 1) / kMaxPieces, piece_min_groups());
        for (uint64_t g0 = 0; g0 < groups; g0 += per) begin[pieces++] = g0; // (none is empty)
    } else {
        const std::vector<uint32_t> &w = piece_schedule();
        uint64_t sum = 0, acc = 0;
        for (uint32_t x : w) sum += x;
        for (size_t i = 0; i < w.size() && pieces < kMaxPieces; ++i) {
            const uint64_t g0 = groups * acc / sum;
            if (pieces == 0 || g0 > begin[pieces - 1]) begin[pieces++] = g0;
            acc += w[i];
        }
    }
    begin[pieces] = groups;
    // coefficient bands in front of the pieces: only where a piece can begin with an MCU row (the groups of 192 blocks and
    // the MCU rows must share boundaries: 4:2:0 widths that are multiples of 512, 4:4:4 of 512, gray of 1536)
    uint64_t groups_per_row = 0;
    if (src) {
        const uint32_t unit = (!src->g->gray && src->g->s420) ? 16u : 8u, units_x = (src->o->width + unit - 1) / unit;
        const uint64_t row_blocks = static_cast<uint64_t>(units_x) * j.a.blocks_per_mcu;
        if (row_blocks % kGroupBlocks == 0) {
            groups_per_row = row_blocks / kGroupBlocks;
            uint32_t kept = 1; // (piece 0 begins at row 0)
            for (uint32_t k = 1; k < pieces; ++k) {
                const uint64_t g0 = begin[k] / groups_per_row * groups_per_row;
                if (g0 > begin[kept - 1]) begin[kept++] = g0;
            }
            pieces = kept;
            begin[pieces] = groups;
        } else { // (no common boundaries: the whole image first)
            if (from_host) {
                const size_t px_bytes = static_cast<size_t>(src->o->width) * src->o->height * (src->g->gray ? 1 : 3);
                HIP_TRY(hipMemcpyAsync(const_cast<void *>(src->d_px), src->host_px, px_bytes, hipMemcpyHostToDevice, stream));
            }
            const int rc = coeffs_rows(c, src->d_px, *src->o, *src->g, stream, src->dy, src->dcb, src->dcr, 0, 0);
            if (rc) return rc;
        }
    }
    const bool upload_bands = from_host && groups_per_row != 0;
    if (upload_bands) {
        if (!c.upload_stream) HIP_TRY(hipStreamCreateWithFlags(&c.upload_stream, hipStreamNonBlocking));
        while (c.band_up.size() < pieces) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c.band_up.push_back(e);
        }
    }
    if (!c.copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking));
    while (c.piece_done.size() < pieces) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c.piece_done.push_back(e);
    }
    HIP_TRY(c.e_chain.reserve(2 * (kMaxPieces + 1) * 8));
    unsigned long long *bits_chain = c.e_chain.as<unsigned long long>(), *out_chain = bits_chain + kMaxPieces + 1;
    const size_t out_cap = std::max<size_t>(j.stream_cap / 4, 4096);
    HIP_TRY(c.e_out.reserve(out_cap));
    // every piece has its own part of the stream buffer (a block has at most 209 bytes; 64 bytes of slack per piece)
    struct Piece { uint64_t first_block, blocks, tiles; uint32_t *stream; };
    Piece pc[kMaxPieces];
    for (uint32_t k = 0; k < pieces; ++k) {
        pc[k].first_block = std::min<uint64_t>(j.n, begin[k] * kGroupBlocks);
        pc[k].blocks = std::min<uint64_t>(j.n, begin[k + 1] * kGroupBlocks) - pc[k].first_block;
        pc[k].tiles = pd::stuff_tiles(pc[k].blocks * 64 + 4096);
    }
    HIP_TRY(c.e_stream.reserve(j.stream_cap + 80 * kMaxPieces));
    for (uint32_t k = 0; k < pieces; ++k) pc[k].stream = c.e_stream.as<uint32_t>() + (pc[k].first_block * 209 + 64 * k + 15) / 16 * 4;
    const size_t state_words = pd::fused_code_state_words(j.n);
    Stopwatch sw;
    uint64_t done = 0; // bytes of the scan that are on their way to the host
    bool redo = false, gave_up = false;
    uint32_t next_out = 0; // first piece whose bytes have not been sent yet
    // piece k has been coded (its event has fired): its bytes onto the copy stream, unless something went wrong
    Stopwatch sw_send; // (send_piece may run on the helper thread: a stopwatch of its own)
    auto send_piece = [&](uint32_t k) -> int {
        if (redo) return PIXO_OK; // (everything that was enqueued is still waited for)
        const uint64_t *mail = c.h_totals + 4 * k;
        if (mail[3]) { redo = gave_up = true; return PIXO_OK; } // (a kernel of this piece gave up waiting: the later pieces follow suit)
        const uint64_t stream_bits = mail[0], packed = k + 1 != pieces ? stream_bits / 8 : (stream_bits + 7) / 8;
        if (pd::stuff_tiles(packed) > pc[k].tiles) { redo = true; return PIXO_OK; } // (the stuffing grid was a guess: this piece is not complete)
        const uint64_t bytes = mail[1];
        if (done + bytes > c.e_out.cap || done + bytes > dst_cap) { redo = true; return PIXO_OK; }
        if (bytes) HIP_TRY(hipMemcpyAsync(dst + done, c.e_out.as<uint8_t>() + done, bytes, hipMemcpyDeviceToHost, c.copy_stream));
        done += bytes;
        sw_send.lap("  copy enqueued");
        return PIXO_OK;
    };
    // piece k: its MCU rows through the coefficient kernel (where the pieces follow the rows), then the two entropy kernels
    auto launch_piece = [&](uint32_t k) -> int {
        pd::ScanArgs a = j.a;
        a.nblocks = pc[k].blocks;
        a.pad_last = k + 1 == pieces ? 1u : 0u;
        const pd::ScanPiece piece{pc[k].first_block, k, bits_chain, k ? pc[k - 1].stream : nullptr};
        unsigned long long *mail = reinterpret_cast<unsigned long long *>(c.h_totals) + 4 * k;
        const bool zero = c.code_state_zero_words >= state_words;
        c.code_state_zero_words = 0;
        if (groups_per_row) {
            const uint32_t row0 = static_cast<uint32_t>(begin[k] / groups_per_row);
            const uint32_t rows = k + 1 == pieces ? 0u : static_cast<uint32_t>(begin[k + 1] / groups_per_row) - row0;
            if (upload_bands) HIP_TRY(hipStreamWaitEvent(stream, c.band_up[k], 0)); // (its pixels arrive on the upload stream)
            const int rc = coeffs_rows(c, src->d_px, *src->o, *src->g, stream, src->dy, src->dcb, src->dcr, row0, rows);
            if (rc) return rc;
        }
        HIP_TRY(pd::launch_scan_code(a, c.e_code_state.as<unsigned long long>(), zero, pc[k].stream, c.e_stuff_state.as<unsigned long long>(),
                                     pd::fused_stuff_state_words(j.stream_cap), mail, stream, &piece, nullptr, debug().spin_budget));
        HIP_TRY(pd::launch_stuff_fused(pc[k].stream, c.e_code_state.as<unsigned long long>(), state_words, 0, /*band=*/k + 1 != pieces,
                                       j.stream_cap, 0, pc[k].tiles, c.e_stuff_state.as<unsigned long long>(), /*state_is_zero=*/true,
                                       c.e_out.as<uint8_t>(), c.e_out.cap, mail, stream, out_chain, k, nullptr, debug().spin_budget));
        c.code_state_zero_words = state_words;
        HIP_TRY(hipEventRecord(c.piece_done[k], stream));
        return PIXO_OK;
    };
    if (upload_bands) {
        // Pixels from the host.  A copy from pageable memory keeps the calling thread inside the runtime until the bytes have
        // moved, and whatever else that thread has to do in between — launches, events — is time the link stands still.  So
        // THIS thread does nothing but enqueue the bands' uploads back to back; the context's helper thread follows it band
        // by band (`uploaded`): launches the band's kernels behind its upload event and sends the piece before it — coded
        // while this band travelled — back over the other direction of the link.
        struct Progress { // bands uploaded so far, handed from the uploading thread to the helper
            std::mutex m;
            std::condition_variable cv;
            uint32_t n = 0;
            void set(uint32_t v) { { std::lock_guard<std::mutex> lock(m); if (v > n) n = v; } cv.notify_all(); }
            void wait_beyond(uint32_t k) { std::unique_lock<std::mutex> lock(m); cv.wait(lock, [&] { return n > k; }); }
        } uploaded;
        int helper_rc = PIXO_OK;
        std::string helper_error; // (the helper's failures set ITS thread's message: carried over to the caller's below)
        if (!c.helper) c.helper = new CopyHelper;
        const int device = c.device;
        c.helper->start([&, device] {
            (void)hipSetDevice(device);
            auto body = [&]() -> int {
                for (uint32_t k = 0; k < pieces; ++k) {
                    uploaded.wait_beyond(k);
                    int rc = launch_piece(k);
                    if (rc) return rc;
                    if (k) {
                        HIP_TRY(hipEventSynchronize(c.piece_done[k - 1]));
                        if ((rc = send_piece(k - 1))) return rc;
                    }
                }
                HIP_TRY(hipEventSynchronize(c.piece_done[pieces - 1]));
                const int rc = send_piece(pieces - 1);
                if (rc) return rc;
                HIP_TRY(hipStreamSynchronize(c.copy_stream));
                return PIXO_OK;
            };
            helper_rc = body();
            if (helper_rc) helper_error = t_error;
        });
        // (whatever happens below, the helper must have finished before this frame's variables go away)
        struct HelperGuard { Context &c; Progress &up; uint32_t n; ~HelperGuard() { up.set(n); c.helper->wait(); } } guard{c, uploaded, pieces};
        const uint32_t unit = (!src->g->gray && src->g->s420) ? 16u : 8u;
        const size_t row_bytes = static_cast<size_t>(src->o->width) * (src->g->gray ? 1 : 3);
        for (uint32_t k = 0; k < pieces; ++k) {
            const size_t r0 = static_cast<size_t>(begin[k] / groups_per_row), r1 = k + 1 == pieces ? ~size_t{0} : static_cast<size_t>(begin[k + 1] / groups_per_row);
            const size_t y0 = std::min<size_t>(src->o->height, r0 * unit), y1 = k + 1 == pieces ? src->o->height : std::min<size_t>(src->o->height, r1 * unit);
            if (y1 > y0)
                HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(const_cast<void *>(src->d_px)) + y0 * row_bytes, src->host_px + y0 * row_bytes,
                                       (y1 - y0) * row_bytes, hipMemcpyHostToDevice, c.upload_stream));
            HIP_TRY(hipEventRecord(c.band_up[k], c.upload_stream));
            uploaded.set(k + 1);
        }
        sw.lap("  bands uploaded");
        c.helper->wait();
        if (helper_rc) return helper_rc > 0 ? helper_rc : fail(helper_rc, helper_error); // (pixo_hip_last_error() is per thread)
    } else {
        for (uint32_t k = 0; k < pieces; ++k) {
            const int rc = launch_piece(k);
            if (rc) return rc;
        }
        sw.lap("  pieces enqueued");
        for (; next_out < pieces; ++next_out) {
            HIP_TRY(hipEventSynchronize(c.piece_done[next_out]));
            sw.lap("  piece coded");
            const int rc = send_piece(next_out);
            if (rc) return rc;
        }
        HIP_TRY(hipStreamSynchronize(c.copy_stream));
    }
    sw.lap("  copies done");
    if (gave_up) return scan_retry_multipass(c);
    if (redo) return 1;
    *scan_bytes = done;
    return PIXO_OK;
}

// Device coefficient tuple -> whole file in the context's PINNED host buffer (headers written by
// the host, entropy-coded segment by the kernels of jpeg_entropy.hip and copied straight behind
// them).  Pinned on purpose: a device-to-host copy into fresh pageable memory makes the runtime
// pin those pages first, which costs 10-25 ms for an 11 MB file every time the address changes.
// batch > 1 (standard tables, no restart markers): the tuples of `batch` equal images back to back;
// every image is a byte-aligned segment of ONE packed stream.  Then *file = headers (once) followed by
// all the entropy-coded segments, and image_starts[i] (batch + 1 entries) are their offsets behind the
// headers; no EOI is written.
static int device_entropy_to_pinned_once(Context &c, const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
                                         const pixo_host::Geometry &g, hipStream_t stream, const uint8_t **file, size_t *file_len,
                                         uint32_t batch, std::vector<uint64_t> *image_starts, size_t *header_len,
                                         uint8_t *dest, size_t dest_cap, bool *own_malloc, const PixelSource *src, bool *tuple_done,
                                         std::vector<uint8_t> *head_out, uint32_t seg_gap, bool *gaps_left)
{ // src != null: the tuple (dy, dcb, dcr = src's) has not been computed yet, see PixelSource.
  // dest != null: the file goes straight into the caller's storage (no pinned intermediate); when it does not
  // fit, *file_len says how much is needed and nothing is copied (PIXO_ERR_BUFFER_TOO_SMALL).
  // own_malloc != null (and no dest): the caller wants the file in malloc'd memory it will own — once the size is known
  // the block is allocated and the device-to-host copy goes straight into it (*own_malloc = true, *file = the block);
  // the copy into pageable memory runs at the link's rate, and what it saves is the second pass over the file from
  // the pinned buffer (tools/ubench/upload.cpp: 0.21 ms + a warm 11 MB memcpy, or 1.30 against 1.38 ms for new pages).
  // *own_malloc stays false when the file was assembled in the pinned buffer after all (a scan coded in pieces).
    if (own_malloc) *own_malloc = false;
    if (!src) *tuple_done = true;
    namespace pd = pixo_dev;
    Stopwatch sw;
    ScanJob j;
    j.seg_gap = batch > 1 ? seg_gap : 0;
    int rc = scan_begin(c, j, dy, dcb, dcr, o, g, batch, nullptr);
    if (rc) return rc;
    if (gaps_left) *gaps_left = j.segmented && j.seg.marker_bytes == seg_gap && seg_gap != 0;
    sw.lap("  reserve");
    std::vector<uint8_t> head;
    // a large scan: in pieces, the file leaving for the host while the rest is still being coded — into the context's
    // pinned buffer, or into the caller's storage if that can hold any file the stuffing grids are sized for (a smaller
    // one might not fit the file, and then nothing may have been written to it: one piece, size first)
    const size_t likely_most = 1024 + static_cast<size_t>(j.n) * 64 + 8192;
    // A medium scan in pieces only pays when the file is large (a 0.3 MB file of a smooth 4096x4096 image: 0.12 ms in one
    // piece, more in two): the context remembers the bytes per block of its last scan and cuts the next one only when that
    // was 12 or more (a stream of similar images; the first one is coded in one piece).
    const uint64_t scan_groups = (j.n + 191) / 192;
    const bool large = scan_groups >= 2 * piece_min_groups();
    const bool medium = !large && scan_groups >= piece_medium_groups() && (c.packed_per_block >= 12 || piece_medium_forced());
    // Pixels still in host memory (pixo_hip_jpeg_encode / _encode_into): their way over PCIe is most of the call.  From 96 MB
    // of pixels on (8192x4096) the image is uploaded in bands, each band transformed and coded while the next one travels,
    // coded pieces on their way back meanwhile (device_entropy_pieces): 16384x16384 17.7 -> 15.1 ms, which is the upload
    // alone at 53 GB/s.  Below that the two extra threads' hand-offs cost what the overlap gains (4096x4096: 1.18 ms either
    // way, of which 0.95 are the upload; profiles/r03_host_pipeline.txt).  Whatever path is taken, the pixels are uploaded once.
    bool host_px_pending = src && src->host_px;
    const bool host_bands = host_px_pending && static_cast<uint64_t>(o.width) * o.height * (g.gray ? 1 : 3) >= (uint64_t{debug().bands_upload_min_mb} << 20) &&
                            !debug().no_bands_upload;
    auto upload_all = [&]() -> int {
        if (!host_px_pending) return PIXO_OK;
        const size_t px_bytes = static_cast<size_t>(o.width) * o.height * (g.gray ? 1 : 3);
        HIP_TRY(hipMemcpyAsync(const_cast<void *>(src->d_px), src->host_px, px_bytes, hipMemcpyHostToDevice, stream));
        host_px_pending = false;
        return PIXO_OK;
    };
    // (Not for a caller that wants a malloc'd block of its own: the block would have to be allocated before the size is
    // known — 64 bytes per block, cut to size afterwards — and a block of a new size is new pages every call, which the
    // device-to-host copy has to fault in and pin: 20 ms instead of 0.7 for the 4096x4096 noise image.  One piece, the
    // exact size, recycled by malloc.)
    // Round 5: pixels that have not been transformed yet go through the fused pixel -> bit stream kernel (jpeg_pixels_code.hip)
    // where that kernel serves the job — one piece: the whole scan is coded ~50 us after the call began, which is where the
    // first of a medium scan's pieces used to be.  Large scans and host pixels in bands keep the pieces (their PCIe time is
    // what the pieces hide); their bands run coefficient kernel + scan_code as before.
    // (A stream of files of more than 30 bytes per block — 4:2:0 above 5.6 bit/px: photographs at q = 100, noise at q >= 90 — would run the
    // fused kernel's two-pass form for groups of several rounds, 25-50 % behind the two-kernel form: the context's last file decides,
    // as it does for the pieces.  profiles/r06_long_groups_chain.txt)
    const bool dense_stream = batch == 1 && c.last_scan_blocks && c.last_scan_bytes > 30 * c.last_scan_blocks && !debug().fused_batch;
    const bool from_pixels = src && !dense_stream && pixels_code_usable(j, o, g, batch);
    if (from_pixels && batch > 1 && gaps_left) *gaps_left = j.seg_gap == seg_gap && seg_gap != 0; // (the fused kernel leaves any gap between its segments)
    // Second session of round 6: a LARGE scan from device pixels takes the fused kernel too when its stores can go straight to where the file
    // is wanted (the library's pinned buffer, or storage of the caller's the GPU can write) — one kernel whose groups finish one after
    // the other IS a pipeline of coding and PCIe: 4096x4096 4:4:4 photo 181 -> 121 us, gradient 144 -> 88, noise 534 -> 498;
    // 8192x8192 4:2:0 332 -> 261 / 264 -> 152 / 928 -> 916 (tools/large_scan_paths.py, profiles/r06_large_scans_one_kernel.txt).  Plain
    // malloc'd destinations keep the pieces (their copy engine overlaps the coding), host pixels in bands as well.
    bool fused_direct_possible = false;
    if (from_pixels && batch == 1 && large && !host_bands && !debug().no_direct_small) {
        if (!dest) fused_direct_possible = true;
        else {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, dest) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) fused_direct_possible = true;
            else (void)hipGetLastError();
        }
    }
    if (j.fused && !j.segmented && batch == 1 && pieces_enabled() && !direct_host_stores() && ((large && !fused_direct_possible) || (medium && !from_pixels) || host_bands) &&
        (!dest || dest_cap >= likely_most) && (host_bands || !(own_malloc && !dest))) {
        PixelSource device_src; // (the same source once the pixels are on the device)
        if (src && o.optimize_huffman) { // (the statistics need the whole tuple)
            if ((rc = upload_all())) return rc;
            if ((rc = coeffs_rows(c, src->d_px, o, g, stream, src->dy, src->dcb, src->dcr, 0, 0))) return rc;
            src = nullptr;
        } else if (src && src->host_px && !host_bands) { // (device pieces of a scan whose pixels come in one copy)
            if ((rc = upload_all())) return rc;
            device_src = *src;
            device_src.host_px = nullptr;
            src = &device_src;
        }
        if ((rc = scan_tables(c, j, o, g, stream, nullptr))) return rc;
        pixo_host::file_headers(head, o, j.h);
        const size_t hdr = head.size();
        uint8_t *buf = dest;
        size_t cap = dest_cap;
        if (buf) advise_huge(buf, std::min(cap, likely_most));
        if (!buf) {
            if ((rc = c.reserve_hfile(likely_most))) return rc;
            buf = c.h_file;
            cap = c.hfile_cap;
        }
        uint64_t scan_bytes = 0;
        rc = device_entropy_pieces(c, j, stream, buf + hdr, cap - hdr - 2, &scan_bytes, src);
        src = nullptr; // (the tuple is complete now, whatever happened)
        host_px_pending = false;
        *tuple_done = true;
        sw.lap("code+stuff+copy (pieces)");
        if (rc < 0 || rc == kRetryMultipass) return rc;
        if (rc == 0) {
            c.packed_per_block = static_cast<uint32_t>(scan_bytes / (j.n ? j.n : 1));
            c.last_scan_bytes = scan_bytes; c.last_scan_blocks = j.n;
            const size_t total = hdr + scan_bytes + 2;
            std::memcpy(buf, head.data(), hdr);
            buf[hdr + scan_bytes] = 0xFF; // EOI
            buf[hdr + scan_bytes + 1] = 0xD9;
            *file = buf;
            *file_len = total;
            if (header_len) *header_len = hdr;
            return PIXO_OK;
        }
        c.code_state_zero_words = 0; // (rc == 1: start over in one piece, below)
    }
    if ((rc = upload_all())) return rc;
    const bool fuse_now = from_pixels && src; // (src is null once a pieces attempt above has computed the tuple)
    if (fuse_now) { // pixels -> finished scan(s) in ONE kernel (below); the tuple is never written (a retry with the multi-pass kernels computes it)
    } else {
        if (src && batch > 1) { // (a batch's tuple: every plane of all images back to back — src->dy / dcb / dcr point into that layout)
            const float *qt_all = nullptr;
            if ((rc = device_tables(c.device, &qt_all))) return rc;
            HIP_TRY(pd::launch_jpeg_coeffs(src->d_px, o.width, o.height, g.gray, g.s420, batch, src->dy, g.gray ? nullptr : src->dcb,
                                           g.gray ? nullptr : src->dcr, qt_all + (o.quality - 1) * pixo_host::kDeviceQtFloats, stream));
        } else if (src && (rc = coeffs_rows(c, src->d_px, o, g, stream, src->dy, src->dcb, src->dcr, 0, 0))) return rc; // one piece: the whole image first
        *tuple_done = true;
    }
    if (j.fused || fuse_now) { // code + stuff back to back, one read-back
        if (fuse_now && o.optimize_huffman) j.count_px = src->d_px; // (optimised tables: the statistics from the pixels as well — scan_tables)
        if ((rc = fuse_now ? scan_tables(c, j, o, g, stream, nullptr) : scan_lengths(c, j, o, g, stream, nullptr, /*wait=*/false))) return rc;
        // One image into host memory the GPU can write — the context's pinned file buffer, or storage of the caller's
        // that is pinned / registered: the stuffing kernel stores straight into it, behind the place of the headers.
        // Small files take that way by default: the stuffing kernel's stores ARE the transfer, and the call has one wait instead
        // of wait + copy + wait — 64x64 62 -> 42 us, 512x512 noise 65 -> 54 us (device pixels -> pinned), any smooth 1080p image
        // 59 -> 48 us; a 1.4 MB file (1080p noise) is where the copy engine wins again (profiles/r04_small_latency.txt).  "Small" =
        // at most 32,768 blocks (1024x1365 px at 4:2:0), or a file predicted below 768 KB from the bytes per block of this
        // context's last scan.  Large files: the debug switch `direct_stores` (slower, profiles/r02_direct_host_stores.txt).
        constexpr uint64_t kDirectBlocks = 32768, kDirectBytes = 768u << 10;
        const bool small_file = !debug().no_direct_small &&
                                (j.n <= kDirectBlocks ||
                                 (c.last_scan_blocks && static_cast<double>(j.n) * static_cast<double>(c.last_scan_bytes) / static_cast<double>(c.last_scan_blocks) <= kDirectBytes));
        HostTarget target;
        bool direct = false;
        // The one-kernel form (round 5) stores directly at EVERY size: its groups finish one after the other, so the stores of the
        // early ones cross PCIe while the late ones are still coding — 4096x4096 photo-like content 0.129 -> 0.112 ms, noise
        // (11 MB, PCIe-bound either way) 0.291 -> 0.283 ms against kernel + copy engine (profiles/r05_whole_file.txt).
        if (batch == 1 && !j.segmented && (direct_host_stores() || small_file || (fuse_now && !debug().no_direct_small))) {
            pixo_host::file_headers(head, o, j.h); // (the tables are known since scan_lengths)
            if (!dest) {
                target.grow = true;
                target.before = head.size();
                target.after = 2;
                direct = true;
            } else if (dest_cap > head.size() + 2) {
                hipPointerAttribute_t at;
                if (hipPointerGetAttributes(&at, dest) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) {
                    target.p = static_cast<uint8_t *>(at.devicePointer) + head.size();
                    target.cap = dest_cap - head.size() - 2;
                    direct = true;
                } else {
                    (void)hipGetLastError(); // (plain malloc'd memory: not an error, the copy below handles it)
                }
            }
        }
        if (fuse_now) rc = scan_from_pixels(c, j, o, g, stream, src->d_px, direct ? &target : nullptr, /*wait=*/true, batch);
        else rc = scan_stuff_fused(c, j, stream, 0, nullptr, nullptr, nullptr, /*chained=*/true, direct ? &target : nullptr);
        if (rc) return rc;
        sw.lap("code+stuff (fused)");
        if (direct) {
            const size_t hdr = head.size(), total = hdr + j.scan_bytes + 2;
            if (dest && j.scan_bytes > target.cap) {
                *file_len = total;
                return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(total) + " bytes");
            }
            if (j.n) { c.packed_per_block = static_cast<uint32_t>(j.scan_bytes / j.n); c.last_scan_bytes = j.scan_bytes; c.last_scan_blocks = j.n; }
            uint8_t *buf = dest ? dest : c.h_file;
            std::memcpy(buf, head.data(), hdr);
            buf[hdr + j.scan_bytes] = 0xFF; // EOI
            buf[hdr + j.scan_bytes + 1] = 0xD9;
            *file = buf;
            *file_len = total;
            if (header_len) *header_len = hdr;
            return PIXO_OK;
        }
    } else {
        if ((rc = scan_lengths(c, j, o, g, stream, nullptr))) return rc;
        sw.lap("tables+lengths+scan");
        if ((rc = scan_pack(c, j, stream))) return rc;
        sw.lap("memset+pack+ff census");
    }
    const uint64_t scan_bytes = j.scan_bytes;
    if (batch == 1 && j.n) { c.packed_per_block = static_cast<uint32_t>(scan_bytes / j.n); c.last_scan_bytes = scan_bytes; c.last_scan_blocks = j.n; }
    if (batch > 1 && (j.segmented || j.pc_seg)) { // the stuffing kernel / the fused kernel left every image's end in the pinned mailbox
        image_starts->assign(batch + 1, 0); // (h_segs[i]: where image i's bytes end; the next image begins behind the gap)
        for (uint32_t i = 0; i < batch; ++i) (*image_starts)[i + 1] = c.h_segs[i] + (i + 1 < batch ? j.seg.marker_bytes : 0);
    } else if (batch > 1) { // where every image's segment begins in the stuffed stream (reuses the seg_bytes buffer: 8 B/entry)
        HIP_TRY(c.e_seg_bytes.reserve(j.nseg * 8));
        HIP_TRY(pd::launch_segment_out_offsets(j.plan, j.nbytes, c.e_stream.as<uint32_t>(), c.e_tile_base.as<uint64_t>(),
                                               c.e_seg_bytes.as<uint64_t>(), stream));
        image_starts->assign(batch + 1, 0);
        HIP_TRY(hipMemcpyAsync(image_starts->data(), c.e_seg_bytes.p, j.nseg * 8, hipMemcpyDeviceToHost, stream));
        (*image_starts)[batch] = scan_bytes;
    }
    head.clear();
    pixo_host::file_headers(head, o, j.h);
    const size_t hdr = head.size(), total = hdr + scan_bytes + 2;
    if (head_out) { // the caller delivers the bytes itself from c.e_out (a batch: every file to its final place)
        if (batch > 1 && !j.segmented && !j.pc_seg) HIP_TRY(hipStreamSynchronize(stream)); // (image_starts is being copied)
        *head_out = head;
        *file = nullptr;
        *file_len = static_cast<size_t>(scan_bytes);
        if (header_len) *header_len = hdr;
        return PIXO_OK;
    }
    uint8_t *buf = dest;
    bool mine = false;
    if (dest) {
        if (total > dest_cap) {
            *file_len = total;
            return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(total) + " bytes");
        }
    } else if (own_malloc && batch == 1) {
        buf = alloc_file(total);
        if (!buf) return fail(PIXO_ERR_COMPRESSION, "Compression error: out of host memory");
        mine = true;
    } else {
        if ((rc = c.reserve_hfile(total))) return rc;
        buf = c.h_file;
    }
    std::memcpy(buf, head.data(), hdr);
    hipError_t ce = hipMemcpyAsync(buf + hdr, c.e_out.p, scan_bytes, hipMemcpyDeviceToHost, stream);
    if (ce == hipSuccess) ce = hipStreamSynchronize(stream);
    if (ce != hipSuccess) {
        if (mine) std::free(buf);
        return hip_fail(ce, "device-to-host copy of the file");
    }
    if (mine) *own_malloc = true;
    buf[hdr + scan_bytes] = 0xFF; // EOI (of the only image; batches append it per file)
    buf[hdr + scan_bytes + 1] = 0xD9;
    *file = buf;
    *file_len = total;
    if (header_len) *header_len = hdr;
    sw.lap("stuff+copy to host");
    return PIXO_OK;
}

int device_entropy_to_pinned(Context &c, const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
                             const pixo_host::Geometry &g, hipStream_t stream, const uint8_t **file, size_t *file_len,
                             uint32_t batch, std::vector<uint64_t> *image_starts, size_t *header_len,
                             uint8_t *dest, size_t dest_cap, bool *own_malloc, const PixelSource *src, std::vector<uint8_t> *head_out,
                             uint32_t seg_gap, bool *gaps_left)
{
    bool tuple_done = false;
    int rc = device_entropy_to_pinned_once(c, dy, dcb, dcr, o, g, stream, file, file_len, batch, image_starts, header_len, dest, dest_cap,
                                           own_malloc, src, &tuple_done, head_out, seg_gap, gaps_left);
    if (rc != kRetryMultipass) return rc;
    // a single-pass kernel gave up waiting (its waits are bounded): the same scan with the multi-pass kernels
    RetryMultipass scope;
    rc = device_entropy_to_pinned_once(c, dy, dcb, dcr, o, g, stream, file, file_len, batch, image_starts, header_len, dest, dest_cap,
                                       own_malloc, tuple_done ? nullptr : src, &tuple_done, head_out, seg_gap, gaps_left);
    return rc == kRetryMultipass ? fail(PIXO_ERR_COMPRESSION, "Compression error: the entropy kernels could not make progress") : rc;
}

// Copies into FRESH host memory are page-fault bound (one core maps and fills a few GB/s of new pages):
// above a few MB the bytes are spread over a handful of threads (PIXO_HIP_COPY_THREADS, default 8; 1 = none).
inline unsigned copy_threads() { return debug().copy_threads; }
// A block the caller will own and give back with pixo_hip_free.  Small files: malloc — glibc serves blocks up to its
// (dynamic, <= 32 MB) mmap threshold from recycled heap pages.  Larger ones would be fresh mappings every time, and a
// fresh mapping costs a page fault per 4 KiB when it is first written: for the 178 MB file of a 16384x16384 image that is
// several times the file's whole way over PCIe, and more threads do not help much (the faults serialise on the address
// space; tools/ubench/fresh_pages.cpp).  So pixo_hip_free keeps up to two large blocks instead of unmapping them and the
// next large file reuses one: its pages are resident.  pixo_hip_trim releases them.
constexpr size_t kLargeBlock = size_t{24} << 20;
struct BlockCache {
    std::mutex m;
    struct Entry { void *p; size_t cap; };
    std::vector<Entry> kept;
    static constexpr size_t kMaxKept = 2, kMaxBytes = size_t{1} << 30;
};
BlockCache &block_cache()
{
    static BlockCache *b = new BlockCache;
    return *b;
}
uint8_t *alloc_file(size_t n)
{
    if (n < kLargeBlock || debug().plain_host) return static_cast<uint8_t *>(std::malloc(n ? n : 1));
    {
        BlockCache &bc = block_cache();
        std::lock_guard<std::mutex> lock(bc.m);
        size_t best = bc.kept.size();
        for (size_t i = 0; i < bc.kept.size(); ++i) // smallest block that holds the file and is not absurdly larger
            if (bc.kept[i].cap >= n && bc.kept[i].cap <= 2 * n + kLargeBlock && (best == bc.kept.size() || bc.kept[i].cap < bc.kept[best].cap)) best = i;
        if (best < bc.kept.size()) {
            void *p = bc.kept[best].p;
            bc.kept.erase(bc.kept.begin() + static_cast<long>(best));
            return static_cast<uint8_t *>(p);
        }
    }
    // a fresh block: 2 MiB aligned with a transparent-huge-page hint — filled by big_copy's threads it costs 2.7 ms for 178 MB
    // on the GPU box where plain malloc'd pages cost 15 (one thread: 28); a recycled block 1.7 (tools/ubench/fresh_pages.cpp,
    // profiles/r03_fresh_pages.txt).  Head-room: the next file of about this size fits the recycled block.
    constexpr size_t kHuge = size_t{2} << 20;
    const size_t rounded = (n + n / 8 + kHuge - 1) / kHuge * kHuge;
    void *p = nullptr;
    if (posix_memalign(&p, kHuge, rounded) != 0) return nullptr;
    (void)madvise(p, rounded, MADV_HUGEPAGE);
    return static_cast<uint8_t *>(p);
}
// ---- files of a BATCH that the caller will own (pixo_hip_jpeg_encode_batch_device) -------------------------------
// 64 x 1.4 MB of fresh malloc'd pages are 22,000 page faults: 26 ms for a batch whose kernels take 0.45 and whose bytes cross PCIe
// in 1.7.  glibc does not keep such blocks either (they are mmap'ed or trimmed off the heap's top).  So the blocks of a batch's
// files come from a pool of PINNED host memory that pixo_hip_free gives back to: resident pages, and the device-to-host copy of a
// file lands in the caller's block directly — no staging buffer, no second pass.  To the caller a block is ordinary host memory
// it owns until pixo_hip_free.  At most kMaxTotal bytes; beyond that (or with debug switch plain_host) plain malloc as before.
struct PinnedPool {
    std::mutex m;
    struct Block { void *p; size_t cap; bool used; };
    std::vector<Block> blocks;
    size_t total = 0;
    static constexpr size_t kMaxTotal = size_t{2} << 30, kGrain = size_t{256} << 10;
};
PinnedPool &pinned_pool()
{
    static PinnedPool *pp = new PinnedPool; // (never destroyed: no HIP call in a static destructor)
    return *pp;
}
uint8_t *pool_take(size_t n)
{
    if (debug().plain_host) return nullptr;
    PinnedPool &pp = pinned_pool();
    std::lock_guard<std::mutex> lock(pp.m);
    size_t best = pp.blocks.size();
    for (size_t i = 0; i < pp.blocks.size(); ++i) // the smallest free block that holds the file and is not absurdly larger
        if (!pp.blocks[i].used && pp.blocks[i].cap >= n && pp.blocks[i].cap <= 2 * n + (size_t{1} << 20) &&
            (best == pp.blocks.size() || pp.blocks[i].cap < pp.blocks[best].cap)) best = i;
    if (best < pp.blocks.size()) {
        pp.blocks[best].used = true;
        return static_cast<uint8_t *>(pp.blocks[best].p);
    }
    const size_t cap = (n + n / 8 + PinnedPool::kGrain) / PinnedPool::kGrain * PinnedPool::kGrain; // (head-room: the next batch's file fits)
    if (pp.total + cap > PinnedPool::kMaxTotal) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    pp.blocks.push_back({p, cap, true});
    pp.total += cap;
    return static_cast<uint8_t *>(p);
}
bool pool_give(void *p)
{
    PinnedPool &pp = pinned_pool();
    std::lock_guard<std::mutex> lock(pp.m);
    for (PinnedPool::Block &b : pp.blocks)
        if (b.p == p) { b.used = false; return true; }
    return false;
}
void pool_drain()
{ // (pixo_hip_trim) the blocks nobody holds go back to the driver
    PinnedPool &pp = pinned_pool();
    std::lock_guard<std::mutex> lock(pp.m);
    std::vector<PinnedPool::Block> keep;
    for (const PinnedPool::Block &b : pp.blocks) {
        if (b.used) keep.push_back(b);
        else { (void)hipHostFree(b.p); pp.total -= b.cap; }
    }
    pp.blocks.swap(keep);
}

void free_file(void *p)
{
    if (!p) return;
    if (pool_give(p)) return;
    const size_t cap = debug().plain_host ? 0 : malloc_usable_size(p);
    if (cap >= kLargeBlock) {
        BlockCache &bc = block_cache();
        std::lock_guard<std::mutex> lock(bc.m);
        size_t bytes = cap;
        for (const BlockCache::Entry &e : bc.kept) bytes += e.cap;
        if (bc.kept.size() < BlockCache::kMaxKept && bytes <= BlockCache::kMaxBytes) {
            bc.kept.push_back({p, cap});
            return;
        }
    }
    std::free(p);
}
void drop_kept_blocks()
{
    pool_drain();
    BlockCache &bc = block_cache();
    std::lock_guard<std::mutex> lock(bc.m);
    for (const BlockCache::Entry &e : bc.kept) std::free(e.p);
    bc.kept.clear();
}

// Transparent huge pages for the whole 2 MiB units inside [p, p + n): a hint before the first touch of a large block the
// CALLER allocated (a fresh 178 MB block is 43,000 page faults otherwise: 15 ms over 8 threads against 2.7 with huge pages,
// profiles/r03_fresh_pages.txt; resident pages are not affected).
void advise_huge(void *p, size_t n)
{
    constexpr uintptr_t kHuge = uintptr_t{2} << 20;
    const uintptr_t a = (reinterpret_cast<uintptr_t>(p) + kHuge - 1) & ~(kHuge - 1), b = (reinterpret_cast<uintptr_t>(p) + n) & ~(kHuge - 1);
    if (n >= kLargeBlock && b > a && !debug().plain_host) (void)madvise(reinterpret_cast<void *>(a), b - a, MADV_HUGEPAGE);
}

void big_copy(uint8_t *dst, const uint8_t *src, size_t n)
{
    advise_huge(dst, n);
    constexpr size_t kSlice = size_t{1} << 20;
    const size_t slices = (n + kSlice - 1) / kSlice;
    const unsigned t = static_cast<unsigned>(std::min<size_t>(copy_threads(), slices / 2));
    if (t <= 1) { std::memcpy(dst, src, n); return; }
    run_on_threads(t, [&](unsigned i) {
        const size_t a = slices * i / t * kSlice, b = std::min(n, slices * (i + 1) / t * kSlice);
        if (b > a) std::memcpy(dst + a, src + a, b - a);
    });
}

// ... and into memory the caller owns: a fresh malloc block, or storage it supplied
int deliver(const uint8_t *file, size_t n, uint8_t **out, size_t *out_len)
{
    uint8_t *p = alloc_file(n);
    if (!p) return fail(PIXO_ERR_COMPRESSION, "Compression error: out of host memory");
    big_copy(p, file, n);
    *out = p;
    *out_len = n;
    return PIXO_OK;
}

int device_entropy_to_malloc(Context &c, const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
                             const pixo_host::Geometry &g, hipStream_t stream, uint8_t **out_buf, size_t *out_len)
{
    const uint8_t *file = nullptr;
    size_t n = 0;
    bool own = false;
    int rc = device_entropy_to_pinned(c, dy, dcb, dcr, o, g, stream, &file, &n, 1, nullptr, nullptr, nullptr, 0, &own);
    if (rc) return rc;
    if (own) { // (already in a block of its own)
        *out_buf = const_cast<uint8_t *>(file);
        *out_len = n;
        return PIXO_OK;
    }
    return deliver(file, n, out_buf, out_len);
}

int hand_over(const std::vector<uint8_t> &v, uint8_t **out, size_t *out_len)
{
    uint8_t *p = alloc_file(v.size());
    if (!p) return fail(PIXO_ERR_COMPRESSION, "Compression error: out of host memory");
    big_copy(p, v.data(), v.size());
    *out = p;
    *out_len = v.size();
    return PIXO_OK;
}

int device_tuple_to_malloc(const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
                           const pixo_host::Geometry &g, Context &c, uint8_t **out, size_t *out_len)
{
    if (!debug().host_entropy) {
        if (!o.progressive) return device_entropy_to_malloc(c, dy, dcb, dcr, o, g, c.stream, out, out_len);
        pixo_host::HuffSet h;
        int rc = huffman_for_tuple(dy, dcb, dcr, o, g, c, h);
        if (rc) return rc;
        std::vector<uint8_t> head;
        pixo_host::file_headers(head, o, h);
        const uint8_t *file = nullptr;
        size_t n = 0;
        if ((rc = device_progressive_scans(dy, dcb, dcr, g, h, c, head, &file, &n))) return rc;
        return deliver(file, n, out, out_len);
    }
    // for experiments, the host twin of the scan coders: host code on a copy of the tuple
    const size_t coef_bytes = (g.y_blocks + 2 * g.c_blocks) * 128;
    int rc = c.reserve_hcoef(coef_bytes);
    if (rc) return rc;
    int16_t *hy = static_cast<int16_t *>(c.h_coef), *hcb = hy + g.y_blocks * 64, *hcr = hcb + g.c_blocks * 64;
    HIP_TRY(hipMemcpyAsync(hy, dy, g.y_blocks * 128, hipMemcpyDeviceToHost, c.stream));
    if (g.c_blocks) {
        HIP_TRY(hipMemcpyAsync(hcb, dcb, g.c_blocks * 128, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipMemcpyAsync(hcr, dcr, g.c_blocks * 128, hipMemcpyDeviceToHost, c.stream));
    }
    HIP_TRY(hipStreamSynchronize(c.stream));
    std::vector<uint8_t> v;
    pixo_host::encode_file(hy, hcb, hcr, o, v);
    return hand_over(v, out, out_len);
}

} // namespace pixo_capi
