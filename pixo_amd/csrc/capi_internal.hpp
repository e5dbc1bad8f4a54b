// capi_internal.hpp — what the translation units behind include/pixo_hip.h share (not installed, not part of the ABI).
//
//   context.cpp      error state, per-device table cache, the thread's Context (stream + grow-only buffers), the pool that
//                    outlives threads, device selection, the debug-switch parser
//   scan_job.cpp     coefficient launches on a context; one pass of the device entropy stage in the steps a band needs
//   pieces.cpp       a scan coded in pieces while the file travels; device tuple / pixels -> whole baseline file
//   progressive.cpp  preset 2: trellis tuple, the seven progressive scans
//   jpeg_api.cpp     the extern "C" JPEG entry points
//   bands.cpp        one image over several GPUs: band encoder, splice, pixo_hip_jpeg_encode_multi
//   png_api.cpp      the extern "C" PNG row-filter entry points
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pixo_hip.h"
#include "jpeg_entropy.hpp"
#include "jpeg_host.hpp"
#include "jpeg_kernels.hpp"
#include "jpeg_pixels_code.hpp"
#include "jpeg_scan_block.h" // (the table form of the flat walk: built on the host, see upload_scan_tables)

namespace pixo_capi {

// ---- errors: negative status + thread-local message equal to pixo::Error's Display string (src/error.rs:50-91) --------
extern thread_local std::string t_error;
int fail(int code, const std::string &msg);
int hip_fail(hipError_t e, const char *what); // pixo::Error::CompressionError(String): "Compression error: {msg}"

#define HIP_TRY(expr)                                                  \
    do {                                                               \
        hipError_t e_ = (expr);                                        \
        if (e_ != hipSuccess) return ::pixo_capi::hip_fail(e_, #expr); \
    } while (0)
// A null pointer where the contract wants an object is a caller bug the Rust API cannot express; the C ABI
// answers it with an error instead of a crash.
#define PIXO_REQUIRE(p) do { if (!(p)) return ::pixo_capi::fail(PIXO_ERR_COMPRESSION, "Compression error: null argument '" #p "'"); } while (0)

// ---- debug switches: ONE environment variable, read once ------------------------------------------------------------
// PIXO_HIP_DEBUG="name[=value],name[=value],..." — A/B experiments and tests only; nothing here changes the bytes of a file.
//   trace              per-phase wall times of the device entropy stage on stderr
//   host_entropy       the host twin of the scan coders instead of the device kernels (jpeg_host.cpp)
//   multipass_entropy  the multi-pass entropy kernels (jpeg_entropy.hip) for every scan instead of the single-pass ones
//   direct_stores      the stuffing kernel stores straight into pinned host memory instead of HBM + copy, for files of every size
//   no_direct_small    ... and never, not even for small files (their default since round 4)
//   one_piece          never code a scan in pieces
//   two_kernel_scan    never use the fused pixel -> bit stream kernel (jpeg_pixels_code.hip): coefficient kernel + scan_code as in rounds 2-4
//   fused_batch        batches through the fused kernel whatever the images' width (the default sends batches of images whose 512-pixel
//                      tiles are at least three quarters full through it, every image a segment: scan_job.cpp pixels_code_usable)
//   piece_groups=n     equal pieces of n groups of 192 blocks instead of 2048
//   piece_medium=n     growing pieces from n groups on instead of 1024, whatever the last file's size
//   piece_schedule=a:b:c   their relative sizes (default 1:3)
//   copy_threads=n     threads that copy finished files above 2 MB into fresh host memory (default 8)
//   spin_budget=n      look-back kernels give up waiting after n polls (default 2^20; tests force the fallback with 0)
//   no_bands_upload    host pixels are uploaded in one copy instead of MCU-row bands pipelined with the kernels
//   plain_host         no host-memory policy (for embedders that own theirs): pixo_hip_free returns every block to malloc at once
//                      (no blocks kept for the next large file), fresh blocks are plain malloc, no madvise(MADV_HUGEPAGE) on the
//                      caller's or the library's memory.  Costs what profiles/r03_fresh_pages.txt shows for files of 24 MiB and more.
struct DebugSwitches {
    bool trace = false, host_entropy = false, multipass_entropy = false, direct_stores = false, one_piece = false;
    bool piece_medium_forced = false, no_bands_upload = false, plain_host = false, no_direct_small = false, two_kernel_scan = false, fused_batch = false;
    int trellis_form = 0; // 1 / 2: the trellis search on one / on eight lanes per block whatever the image's size (jpeg_trellis.hpp)
    int coef_form = 0; // 1 / 2: the coefficient kernel's scalar / packed form whatever the launch size (jpeg_kernels.hpp)
    bool no_side_stats = false; // preset 2 on small images: statistics on the context's stream, in front of the search (round 4's order)
    // host pixels are uploaded in bands from this many MiB of pixels on (bands_upload_min_mb=n), in bands of about
    // bands_upload_mb=n MiB.  A 4096x4096 RGB image (48 MiB) in six bands of 8 MiB: noise 1.18 -> 1.14 ms into caller storage,
    // but a smooth image 0.99 -> 1.10 ms and the malloc'ing entry 1.43 -> 1.55: not below 96 MiB (profiles/r03_host_bands_probe.txt)
    uint32_t bands_upload_min_mb = 96;
    uint32_t bands_upload_mb = 12;
    uint64_t piece_groups = 2048, piece_medium = 1024;
    std::vector<uint32_t> piece_schedule{1, 3};
    unsigned copy_threads = 8;
    uint32_t spin_budget = 1u << 20;
    uint32_t batch_parts = 0; // (measurements) sub-batches of pixo_hip_jpeg_encode_batch_device_into, 0 = the library's choice
};
const DebugSwitches &debug();

// ---- per-device table cache: 100 qualities x kDeviceQtFloats floats, uploaded once --------------------------------------
constexpr int kMaxDevices = 64;
int device_tables(int device, const float **out);

// Makes a context's device current for the calling thread for the duration of an entry point and gives
// the caller its own device back afterwards (torch and other HIP users share the thread).
struct DeviceScope {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device); else prev = -1;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// ---- thread-local execution context --------------------------------------------------
struct Context {
    int device = 0;
    bool ready = false;
    hipStream_t stream = nullptr;
    hipEvent_t producer_done = nullptr; // orders the context's stream after the caller's (device-pointer entries)
    hipEvent_t stats_done = nullptr;    // preset 2: the symbol counts of the statistics pass have reached the host (progressive.cpp)
    hipEvent_t side_ready = nullptr;    // preset 2, small images: the second stream may start (the pixels are there)
    void *d_px = nullptr;   size_t px_cap = 0;
    void *d_coef = nullptr; size_t coef_cap = 0;
    void *h_coef = nullptr; size_t hcoef_cap = 0; // pinned
    // device entropy stage (grow-only)
    struct Buf {
        void *p = nullptr; size_t cap = 0;
        hipError_t reserve(size_t n)
        {
            if (n <= cap) return hipSuccess;
            if (p) (void)hipFree(p);
            p = nullptr; cap = 0;
            const size_t want = n + n / 4; // head-room: sizes are data dependent
            hipError_t e = hipMalloc(&p, want);
            if (e == hipSuccess) cap = want;
            return e;
        }
        template <class T> T *as() const { return static_cast<T *>(p); }
    };
    Buf e_tables, e_hist, e_count, e_len, e_off, e_tmp, e_totals, e_stream, e_tile_ff, e_tile_base, e_out, e_seg_bytes, e_seg_off;
    Buf e_code_state, e_stuff_state; // single-pass kernels (jpeg_scan_fused.hip): look-back descriptors, totals
    Buf e_pc_state;                  // the fused pixel -> scan kernel (jpeg_pixels_code.hip): TWO state blocks that alternate — a launch zeroes the
    size_t pc_half_words = 0;        //   block of the launch before it (words per block; 0: nothing is known to be zero)
    int pc_flip = 0;                 //   which block the next launch uses
    Buf e_pc_spill;                  //   where a group of several 6 KiB rounds parks its quantised blocks between the rounds' walks (a buffer of its
                                     //   own: growing d_coef here would free the tuple a caller's retry path still points into)
    Buf e_chain;                     // a scan coded in pieces: bits / bytes of the scan before every piece (device_entropy_pieces)
    Buf e_seams;                     // batch files that stay in HBM: their offsets + the header bytes for batch_seams_kernel
    Buf e_segs;                      // segmented scans (batches, restart intervals): per-segment results of the single-pass kernels
    hipStream_t copy_stream = nullptr; // ... whose bytes travel to the host on this stream while the next piece is coded
    hipStream_t upload_stream = nullptr; // host pixels arrive band by band on this stream while earlier bands are transformed
    struct CopyHelper *helper = nullptr; // ... and a second host thread sends the coded pieces back meanwhile (pieces.cpp; stopped and joined by release())
    std::vector<hipEvent_t> piece_done, band_up;
    uint32_t *h_tables = nullptr; // pinned staging of tables_held for the upload
    uint32_t tables_held[pixo_scan::kScanTableUpload]; bool tables_valid = false; hipStream_t tables_stream = nullptr; // what e_tables holds (no upload when unchanged)
    uint32_t packed_per_block = 0; // bytes per block of the last whole scan this context coded (0: none yet), see device_entropy_to_pinned
    uint64_t last_prog_bytes = 0; // the last progressive file's entropy-coded bytes (small: the next one is stored directly)
    uint64_t last_scan_bytes = 0, last_scan_blocks = 0; // ... exactly (a smooth image is below one byte per block): predicts the next file's size
    uint32_t batch_per_block = 0;  // ... of the last batch (1 + bytes per block; 0: none yet): whether sub-batches pay, jpeg_api.cpp
    size_t code_state_zero_words = 0; // this many words of e_code_state are known to be zero (the stuffing kernel cleans up behind itself)
    Buf p_in, p_out, p_sums, p_scratch; // PNG filter stage
    Buf t_raw, t_trail;                 // progressive + trellis: unquantised DCT blocks (f32), Viterbi back-pointers
    Buf t_plain;                        // preset 2, small images: the plain quantiser's tuple of the statistics pass on the second stream
    Buf g_flags, g_rank, g_by_rank;     // progressive scans: band flags, rank among non-empty blocks and its inverse
    unsigned long long *h_sums = nullptr; size_t hsums_cap = 0; // pinned
    uint64_t *h_totals = nullptr; // pinned, kTotalsWords words: the kernels' mailbox (4 words per piece of a scan)
    static constexpr size_t kTotalsWords = 4 * 32;
    uint64_t *h_segs = nullptr; size_t hsegs_cap = 0; // pinned: per-segment byte offsets of a segmented scan
    uint8_t *h_file = nullptr; size_t hfile_cap = 0; // pinned: the finished file lands here

    int ensure();
    int reserve_px(size_t n);
    int reserve_coef(size_t n);
    int reserve_hcoef(size_t n);
    int reserve_hfile(size_t n);
    int reserve_hsegs(size_t words);
    int ensure_totals();
    size_t held_bytes() const; // device + pinned bytes this context keeps
    void shrink_to(size_t max_buffer_bytes); // releases every buffer larger than this (a parked context keeps the small ones)
    void release(); // everything back to the driver; the context starts over at its next use
};

// Contexts outlive the threads that use them (context.cpp): a thread that ends parks its context in the pool — no HIP
// call in a thread-local destructor — and the next thread that needs one adopts it.
struct ContextPool {
    std::mutex m;
    std::vector<Context *> idle;
    Context *take(int device);
    void give(Context *c); // no HIP calls: may run in a thread-local destructor
    void drain();          // frees every parked context (pixo_hip_trim)
};
ContextPool &pool();
struct ThreadSlot {
    Context *c = nullptr;
    int device = 0; // pixo_hip_set_device
    ~ThreadSlot();
};
extern thread_local ThreadSlot t_slot;
Context &thread_context();

#define PIXO_ON_DEVICE_OF(ctx)                                    \
    ::pixo_capi::DeviceScope device_scope_((ctx).device);         \
    if (device_scope_.err != hipSuccess) return ::pixo_capi::hip_fail(device_scope_.err, "hipSetDevice")

// Device-pointer entry points run on the context's own stream.  What the caller enqueued before the call — on the stream
// it named with pixo_hip_set_producer_stream, by default the NULL stream — is ordered in front of it with an event.
int order_after_producer(Context &c);
int context_on_current_device(Context **out); // binds the thread's context to the HIP device that is current for the caller

struct Stopwatch { // debug switch `trace`: per-phase wall times of the device entropy stage on stderr
    bool on = debug().trace;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pixo_hip] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

void destroy_copy_helper(struct CopyHelper *h); // pieces.cpp (the type is complete only there)

// ---- host memory helpers (pieces.cpp) -------------------------------------------------------------------------------
template <class F> void run_on_threads(unsigned t, F &&body) // body(index) for index in [0, t)
{
    if (t <= 1) { body(0u); return; }
    std::vector<std::thread> workers;
    workers.reserve(t - 1);
    for (unsigned i = 1; i < t; ++i) workers.emplace_back([&body, i] { body(i); });
    body(0u);
    for (auto &w : workers) w.join();
}
void big_copy(uint8_t *dst, const uint8_t *src, size_t n);
void advise_huge(void *p, size_t n);
uint8_t *alloc_file(size_t n); // a block for a finished file that the caller will own: large ones come from the blocks pixo_hip_free kept
void free_file(void *p);       // pixo_hip_free: large blocks are kept (at most two) for the next large file
void drop_kept_blocks();       // pixo_hip_trim
void drop_batch_worker_buffers(); // pixo_hip_trim: the device buffers of pixo_hip_jpeg_encode_batch_multi's worker threads (bands.cpp)
uint8_t *pool_take(size_t n);  // a block of PINNED host memory for a file the caller will own (null: none to be had — malloc instead); back via free_file
int deliver(const uint8_t *file, size_t n, uint8_t **out, size_t *out_len);   // a fresh malloc block the caller owns
int hand_over(const std::vector<uint8_t> &v, uint8_t **out, size_t *out_len);

// ---- coefficient launches on a context (scan_job.cpp) -----------------------------------------------------------------
int coeffs_to_pinned(Context &c, const uint8_t *pixels, const pixo_jpeg_options &o, const pixo_host::Geometry &g,
                     const int16_t **y, const int16_t **cb, const int16_t **cr);
int coeffs_reserve(Context &c, const pixo_host::Geometry &g, int16_t **dy, int16_t **dcb, int16_t **dcr);
int coeffs_rows(Context &c, const void *d_pixels, const pixo_jpeg_options &o, const pixo_host::Geometry &g, hipStream_t stream,
                int16_t *dy, int16_t *dcb, int16_t *dcr, uint32_t row0, uint32_t rows);
int coeffs_on_device(Context &c, const void *d_pixels, const pixo_jpeg_options &o, const pixo_host::Geometry &g, hipStream_t stream,
                     int16_t **dy, int16_t **dcb, int16_t **dcr);
bool scan_has_restart_markers(const pixo_jpeg_options &o, const pixo_host::Geometry &g);

// ---- the device entropy stage, in the steps a caller may need to interleave with exchanges (scan_job.cpp) ------------
// One pass over a coefficient tuple in HBM: a whole image, a batch of images (one byte-aligned segment each), or a
// BAND of a larger image (predictors seeded from the band above, packed at the band's bit offset modulo 8,
// no final padding).
struct ScanJob {
    const void *count_px = nullptr; // optimised tables of a scan the fused kernel codes: the statistics come from these device pixels (scan_tables)
    pixo_dev::ScanArgs a;
    uint64_t n = 0, nseg = 0;
    size_t tmp_blocks = 0, tmp_segs = 0, tmp_tiles = 0;
    pixo_dev::SegmentPlan plan{0, nullptr};
    pixo_host::HuffSet h;
    uint64_t total_bits = 0;
    uint64_t nbytes = 0;     // bytes of the packed stream that get stuffed (a band: its whole bytes only)
    uint64_t scan_bytes = 0; // ... after stuffing, in c.e_out
    bool band = false;
    bool tables_ready = false; // j.h is built and on the device (scan_tables)
    int head_bits = 0;       // band: how many of its first bits share a byte with the band before
    bool fused = false;      // one uninterrupted scan: the two single-pass kernels of jpeg_scan_fused.hip
    bool segmented = false;  // byte-aligned segments (images of a batch, restart intervals) in the single-pass kernels
    bool pc_seg = false;     // ... coded as segments of the fused pixel -> scan kernel (scan_from_pixels): c.h_segs, seg.marker_bytes as for `segmented`
    pixo_dev::SegArgs seg;   // ... their geometry and per-segment arrays (c.e_segs, c.h_segs)
    uint32_t seg_gap = 0;    // set BEFORE scan_begin: bytes a batch wants left free between its images' scans in c.e_out
                             // (headers + EOI: the whole batch then leaves the device in one copy); honoured only by segmented jobs
    size_t stream_cap = 0;   // fused: bytes the packed stream can take at most
    size_t code_state_words = 0; // fused: u64 words of c.e_code_state the code kernel of this job uses (the stuffing kernel zeroes them again)
};
// A step of a job returns this when a single-pass kernel gave up waiting (bounded look-back, jpeg_scan_fused.hip): nothing of
// the job's results is valid; run the job again inside a RetryMultipass scope, which makes scan_begin choose the multi-pass
// kernels.  (Internal: never returned through the C ABI.)
constexpr int kRetryMultipass = 1000;
extern thread_local bool t_force_multipass;
struct RetryMultipass {
    RetryMultipass() { t_force_multipass = true; }
    ~RetryMultipass() { t_force_multipass = false; }
};
int scan_retry_multipass(Context &c);
uint64_t lookback_fallbacks(); // how often that has happened in this process (tests)
int upload_scan_tables(Context &c, const uint32_t (&packed)[pixo_host::kScanTableWords], hipStream_t stream);
int scan_begin(Context &c, ScanJob &j, const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
               const pixo_host::Geometry &g, uint32_t batch, const int16_t *band_seed_dc);
void split_counts(const uint64_t counts[pixo_host::kScanTableWords], uint64_t dc[2][12], uint64_t ac[2][256]);
int scan_count(Context &c, ScanJob &j, hipStream_t stream, uint64_t counts[pixo_host::kScanTableWords]);
int scan_tables(Context &c, ScanJob &j, const pixo_jpeg_options &o, const pixo_host::Geometry &g, hipStream_t stream, const uint64_t *counts);
int scan_lengths(Context &c, ScanJob &j, const pixo_jpeg_options &o, const pixo_host::Geometry &g, hipStream_t stream,
                 const uint64_t *counts, bool wait = true);
// Where the stuffed bytes go when not into the context's device buffer: host memory the GPU can write (pinned), so that
// the kernel's stores ARE the transfer — no second pass over the file, no second synchronisation.
struct HostTarget {
    uint8_t *p = nullptr; // device-visible address of the first stuffed byte
    size_t cap = 0;       // bytes available from there
    bool grow = false;    // p lies in the context's own pinned file buffer: too small = reserve more and repeat
    size_t before = 0, after = 0; // (grow) bytes the file needs in front of / behind the stuffed bytes
};
// The fused pixel -> bit stream kernel (jpeg_pixels_code.hip) in place of coefficient kernel + scan_code for this job?  (one
// RGB image, one uninterrupted scan, tables known without the tuple's statistics)
bool pixels_code_usable(const ScanJob &j, const pixo_jpeg_options &o, const pixo_host::Geometry &g, uint32_t batch);
// ... tables + that kernel: afterwards the finished (stuffed, padded) scan lies in c.e_out — or at `host`, memory of the host
// that the GPU can write — and j.scan_bytes / j.total_bits / j.nbytes say how long it is.  No tuple, no packed stream is written.
// wait = false: only enqueued (measurements); the totals are then in c.h_totals[0..2] once the stream has been synchronised.
int scan_from_pixels(Context &c, ScanJob &j, const pixo_jpeg_options &o, const pixo_host::Geometry &g, hipStream_t stream, const void *d_pixels,
                     HostTarget *host, bool wait = true, uint32_t batch = 1);
int scan_stuff_fused(Context &c, ScanJob &j, hipStream_t stream, uint64_t band_bit_offset, uint32_t *head, int *tail_bits,
                     uint32_t *tail, bool chained = false, HostTarget *host = nullptr);
int scan_pack(Context &c, ScanJob &j, hipStream_t stream, uint64_t band_bit_offset = 0, uint32_t *head = nullptr,
              int *tail_bits = nullptr, uint32_t *tail = nullptr);

// ---- whole baseline files (pieces.cpp) ----------------------------------------------------------------------------
// Pixels whose coefficients have not been computed yet (the tuple's place is reserved): the entropy stage launches the
// coefficient kernel itself — for a scan coded in pieces, band by band in front of each piece.  host_px != null: the
// pixels are still in HOST memory and are uploaded band by band as well (upload_stream), each band's kernels waiting
// only for its own rows.
struct PixelSource {
    const void *d_px;
    const pixo_jpeg_options *o;
    const pixo_host::Geometry *g;
    int16_t *dy, *dcb, *dcr;
    const uint8_t *host_px = nullptr;
};
int device_entropy_to_pinned(Context &c, const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
                             const pixo_host::Geometry &g, hipStream_t stream, const uint8_t **file, size_t *file_len,
                             uint32_t batch = 1, std::vector<uint64_t> *image_starts = nullptr, size_t *header_len = nullptr,
                             uint8_t *dest = nullptr, size_t dest_cap = 0, bool *own_malloc = nullptr, const PixelSource *src = nullptr,
                             std::vector<uint8_t> *head_out = nullptr, uint32_t seg_gap = 0, bool *gaps_left = nullptr);
// (head_out != null: no copy to the host — *head_out receives the file headers, *file_len the bytes of the stuffed scan(s)
// left in c.e_out, image_starts where each image's bytes begin; the caller delivers them.  seg_gap: bytes to leave free in
// c.e_out between consecutive images' scans, *gaps_left says whether that was done — only segmented single-pass jobs can —:
// image_starts then counts the gaps, image i's bytes are [starts[i], starts[i + 1] - gap).)
int device_entropy_to_malloc(Context &c, const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
                             const pixo_host::Geometry &g, hipStream_t stream, uint8_t **out_buf, size_t *out_len);
int device_tuple_to_malloc(const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
                           const pixo_host::Geometry &g, Context &c, uint8_t **out, size_t *out_len);

// ---- preset 2 (progressive.cpp) -----------------------------------------------------------------------------------
int huffman_for_tuple(const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
                      const pixo_host::Geometry &g, Context &c, pixo_host::HuffSet &h);
int device_progressive_scans(const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_host::Geometry &g,
                             const pixo_host::HuffSet &h, Context &c, const std::vector<uint8_t> &head, const uint8_t **file,
                             size_t *file_len, uint8_t *pinned_dest = nullptr, size_t dest_cap = 0);
// (pinned_dest: caller storage the GPU can write — the file is assembled there when it fits, *file == pinned_dest then)
int progressive_to_view(const void *d_pixels, const pixo_jpeg_options &o, const pixo_host::Geometry &g, Context &c,
                        std::vector<uint8_t> &spill, const uint8_t **file, size_t *file_len, uint8_t *pinned_dest = nullptr,
                        size_t dest_cap = 0);

} // namespace pixo_capi
