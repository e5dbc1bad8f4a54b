// capi.cpp — the extern "C" boundary declared in include/pixo_hip.h.
//
// Owns the thread-local device context (HIP stream, grow-only device and pinned host
// buffers) and the per-device quantiser-table cache.  No CPU fallback exists: without a
// usable GPU every compute entry point fails with PIXO_ERR_COMPRESSION and says so.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pixo_hip.h"
#include "jpeg_entropy.hpp"
#include "jpeg_host.hpp"
#include "jpeg_kernels.hpp"
#include "jpeg_scan_block.h" // (the table form of the flat walk: built on the host, see upload_scan_tables)
#include "jpeg_trellis.hpp"
#include "png_filter.hpp"

namespace {

thread_local std::string t_error = "";

int fail(int code, const std::string &msg)
{
    t_error = msg;
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    // pixo::Error::CompressionError(String) Display: "Compression error: {msg}"
    return fail(PIXO_ERR_COMPRESSION, std::string("Compression error: HIP ") + what + ": " +
                                          hipGetErrorString(e));
}

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)

// ---- per-device table cache: 100 qualities x 256 floats, uploaded once -------------
constexpr int kMaxDevices = 64;
std::mutex g_qt_mutex;
float *g_qt[kMaxDevices] = {};

int device_tables(int device, const float **out)
{
    std::lock_guard<std::mutex> lock(g_qt_mutex);
    if (device < 0 || device >= kMaxDevices) return fail(PIXO_ERR_COMPRESSION, "Compression error: bad device index");
    if (!g_qt[device]) {
        std::vector<float> host(100 * pixo_host::kDeviceQtFloats);
        for (int q = 1; q <= 100; ++q) pixo_host::fill_device_qt(static_cast<uint8_t>(q), &host[(q - 1) * pixo_host::kDeviceQtFloats]);
        float *d = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), host.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
        g_qt[device] = d;
    }
    *out = g_qt[device];
    return PIXO_OK;
}

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
    Buf e_chain;                     // a scan coded in pieces: bits / bytes of the scan before every piece (device_entropy_pieces)
    hipStream_t copy_stream = nullptr; // ... whose bytes travel to the host on this stream while the next piece is coded
    std::vector<hipEvent_t> piece_done;
    uint32_t tables_held[pixo_scan::kScanTableUpload]; bool tables_valid = false; hipStream_t tables_stream = nullptr; // what e_tables holds (no upload when unchanged)
    uint32_t packed_per_block = 0; // bytes per block of the last whole scan this context coded (0: none yet), see device_entropy_to_pinned
    size_t code_state_zero_words = 0; // this many words of e_code_state are known to be zero (the stuffing kernel cleans up behind itself)
    Buf p_in, p_out, p_sums, p_scratch; // PNG filter stage
    Buf t_raw, t_trail;                 // progressive + trellis: unquantised DCT blocks (f32), Viterbi back-pointers
    Buf g_flags, g_rank, g_by_rank;     // progressive scans: band flags, rank among non-empty blocks and its inverse
    unsigned long long *h_sums = nullptr; size_t hsums_cap = 0; // pinned
    uint64_t *h_totals = nullptr; // pinned, kTotalsWords words: the kernels' mailbox (4 words per piece of a scan)
    static constexpr size_t kTotalsWords = 4 * 32;
    uint8_t *h_file = nullptr; size_t hfile_cap = 0; // pinned: the finished file lands here
    int reserve_hfile(size_t n)
    {
        if (n > hfile_cap) {
            if (h_file) (void)hipHostFree(h_file);
            h_file = nullptr; hfile_cap = 0;
            const size_t want = n + n / 4;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_file), want, hipHostMallocDefault));
            hfile_cap = want;
        }
        return PIXO_OK;
    }

    int ensure()
    {
        if (ready) return PIXO_OK;
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n == 0)
            return fail(PIXO_ERR_COMPRESSION,
                        "Compression error: no MI355X/HIP device available (pixo_hip has no CPU fallback)");
        if (device < 0 || device >= n) return fail(PIXO_ERR_COMPRESSION, "Compression error: no HIP device " + std::to_string(device));
        DeviceScope on(device);
        if (on.err != hipSuccess) return hip_fail(on.err, "hipSetDevice");
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        ready = true;
        return PIXO_OK;
    }
    int reserve_px(size_t n)
    {
        if (n <= px_cap) return PIXO_OK;
        if (d_px) (void)hipFree(d_px);
        d_px = nullptr; px_cap = 0;
        HIP_TRY(hipMalloc(&d_px, n));
        px_cap = n;
        return PIXO_OK;
    }
    int reserve_coef(size_t n)
    {
        if (n > coef_cap) {
            if (d_coef) (void)hipFree(d_coef);
            d_coef = nullptr; coef_cap = 0;
            HIP_TRY(hipMalloc(&d_coef, n));
            coef_cap = n;
        }
        return PIXO_OK;
    }
    int reserve_hcoef(size_t n)
    {
        if (n > hcoef_cap) {
            if (h_coef) (void)hipHostFree(h_coef);
            h_coef = nullptr; hcoef_cap = 0;
            HIP_TRY(hipHostMalloc(&h_coef, n, hipHostMallocDefault));
            hcoef_cap = n;
        }
        return PIXO_OK;
    }
    void release(); // everything back to the driver; the context starts over at its next use
};
void Context::release()
{
    if (!ready) return;
    DeviceScope on(device);
    if (on.err != hipSuccess) return;
    if (stream) (void)hipStreamSynchronize(stream);
    Buf *bufs[] = {&e_tables, &e_hist, &e_count, &e_len, &e_off, &e_tmp, &e_totals, &e_stream, &e_tile_ff, &e_tile_base, &e_out, &e_seg_bytes,
                   &e_seg_off, &e_code_state, &e_stuff_state, &e_chain, &p_in, &p_out, &p_sums, &p_scratch, &t_raw, &t_trail, &g_flags, &g_rank, &g_by_rank};
    for (Buf *b : bufs) {
        if (b->p) (void)hipFree(b->p);
        b->p = nullptr; b->cap = 0;
    }
    if (d_px) (void)hipFree(d_px);
    if (d_coef) (void)hipFree(d_coef);
    if (h_coef) (void)hipHostFree(h_coef);
    if (h_sums) (void)hipHostFree(h_sums);
    if (h_totals) (void)hipHostFree(h_totals);
    if (h_file) (void)hipHostFree(h_file);
    if (stream) (void)hipStreamDestroy(stream);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    copy_stream = nullptr;
    for (hipEvent_t e : piece_done) (void)hipEventDestroy(e);
    piece_done.clear();
    if (producer_done) (void)hipEventDestroy(producer_done);
    producer_done = nullptr;
    code_state_zero_words = 0;
    tables_valid = false;
    d_px = d_coef = h_coef = nullptr; px_cap = coef_cap = hcoef_cap = 0;
    h_sums = nullptr; hsums_cap = 0; h_totals = nullptr; h_file = nullptr; hfile_cap = 0;
    stream = nullptr; ready = false;
}

// Contexts outlive the threads that use them.  A thread's context must not be torn down by a
// thread-local destructor: those run when the HIP runtime's own per-thread state may already be gone
// (it was created later, so it is destroyed earlier), and — for threads still winding down while main()
// returns — concurrently with the runtime's atexit teardown; hipFree / hipHostFree from there crashed
// (SIGSEGV in amd::Context::svmFree with 16 threads ending at once, profiles/r02_thread_exit_crash.txt).
// So a thread that ends only parks its context here (a mutex and a vector push, no HIP call); the next
// thread that needs one adopts it — buffers, stream and all, which also spares a server with a thread per
// request every hipMalloc — and live threads free what is left over beyond a small reserve.  Nothing is
// destroyed at process exit (the pool is leaked on purpose).
constexpr size_t kIdleContextsKept = 16;
struct ContextPool {
    std::mutex m;
    std::vector<Context *> idle;
    Context *take(int device)
    {
        std::vector<Context *> excess;
        Context *c = nullptr;
        {
            std::lock_guard<std::mutex> lock(m);
            for (size_t i = idle.size(); i-- > 0;)
                if (idle[i]->device == device) { c = idle[i]; idle.erase(idle.begin() + static_cast<long>(i)); break; }
            if (!c && !idle.empty()) { c = idle.back(); idle.pop_back(); }
            while (idle.size() > kIdleContextsKept) { excess.push_back(idle.front()); idle.erase(idle.begin()); }
        }
        for (Context *x : excess) { x->release(); delete x; } // (a live thread: HIP calls are fine here)
        if (!c) c = new Context;
        if (c->device != device) { c->release(); c->device = device; } // rebind: drop the other device's buffers
        return c;
    }
    void give(Context *c) // no HIP calls: may run in a thread-local destructor
    {
        std::lock_guard<std::mutex> lock(m);
        idle.push_back(c);
    }
    void drain() // frees every parked context (pixo_hip_trim)
    {
        std::vector<Context *> all;
        {
            std::lock_guard<std::mutex> lock(m);
            all.swap(idle);
        }
        for (Context *x : all) { x->release(); delete x; }
    }
};
ContextPool &pool()
{
    static ContextPool *p = new ContextPool;
    return *p;
}
struct ThreadSlot {
    Context *c = nullptr;
    int device = 0; // pixo_hip_set_device
    ~ThreadSlot() { if (c) pool().give(c); }
};
thread_local ThreadSlot t_slot;
Context &thread_context()
{
    if (!t_slot.c) t_slot.c = pool().take(t_slot.device);
    return *t_slot.c;
}

#define PIXO_ON_DEVICE_OF(ctx)                      \
    DeviceScope device_scope_((ctx).device);        \
    if (device_scope_.err != hipSuccess) return hip_fail(device_scope_.err, "hipSetDevice")

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

struct Stopwatch { // PIXO_HIP_TRACE=1: per-phase wall times of the device entropy stage on stderr
    bool on = std::getenv("PIXO_HIP_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pixo_hip] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// ---- the device entropy stage, in the steps a caller may need to interleave with exchanges ------------
// One pass of jpeg_entropy.hip over a coefficient tuple in HBM: a whole image, a batch of images (one
// byte-aligned segment each), or a BAND of a larger image (predictors seeded from the band
// above, packed at the band's bit offset modulo 8, no final padding).
struct ScanJob {
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
    size_t stream_cap = 0;   // fused: bytes the packed stream can take at most
};

// PIXO_HIP_OLD_ENTROPY=1: the multi-pass kernels of jpeg_entropy.hip for every scan (A/B runs; they remain the path of
// scans with restart markers, batches and progressive scans)
// The packed Huffman tables of a scan into e_tables — unless they are what the buffer holds already (the standard
// tables, image after image: one small copy less on the stream per file).
int upload_scan_tables(Context &c, const uint32_t (&packed)[pixo_host::kScanTableWords], hipStream_t stream)
{
    if (c.tables_valid && c.tables_stream == stream && std::memcmp(c.tables_held, packed, sizeof packed) == 0) return PIXO_OK;
    c.tables_valid = false;
    std::memcpy(c.tables_held, packed, sizeof packed);
    for (int i = 0; i < pixo_scan::kWalkWords; ++i) // the same tables in the form of the flat walk (jpeg_scan_block.h)
        c.tables_held[pixo_scan::kTableWords + i] = pixo_scan::walk_table_word(packed, i);
    HIP_TRY(hipMemcpyAsync(c.e_tables.p, c.tables_held, sizeof c.tables_held, hipMemcpyHostToDevice, stream));
    c.tables_valid = true;
    c.tables_stream = stream;
    return PIXO_OK;
}

bool direct_host_stores()
{ // PIXO_HIP_DIRECT_STORES=1: the stuffing kernel stores the file straight into pinned host memory instead of into HBM
  // with a copy behind it.  Off by default: kernel stores cross PCIe at 40 GB/s, the copy engine at 53 — 0.36 against
  // 0.32 ms for the 11 MB file of the 4096x4096 noise image (profiles/r02_direct_host_stores.txt); it saves a
  // synchronisation, which only pays for files of a few hundred KB.
    static const bool on = [] { const char *e = std::getenv("PIXO_HIP_DIRECT_STORES"); return e && *e && *e != '0'; }();
    return on;
}

bool old_entropy_forced()
{
    static const bool v = std::getenv("PIXO_HIP_OLD_ENTROPY") != nullptr;
    return v;
}

// Geometry of the pass and every buffer whose size does not depend on the data.
int scan_begin(Context &c, ScanJob &j, const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
               const pixo_host::Geometry &g, uint32_t batch, const int16_t *band_seed_dc)
{
    namespace pd = pixo_dev;
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
    j.fused = j.nseg == 0 && !old_entropy_forced();
    HIP_TRY(c.e_tables.reserve(pixo_scan::kScanTableUpload * 4));
    HIP_TRY(c.e_hist.reserve(pixo_host::kScanTableWords * 8));
    if (j.fused) { // a block has at most 1665 bits: the packed stream has at most n * 209 bytes (+ slack the kernels read into)
        j.stream_cap = static_cast<size_t>(j.n) * 209 + 64;
        HIP_TRY(c.e_stream.reserve(j.stream_cap));
        if (pd::fused_code_state_words(j.n) * 8 > c.e_code_state.cap) c.code_state_zero_words = 0; // (a new buffer)
        HIP_TRY(c.e_code_state.reserve(pd::fused_code_state_words(j.n) * 8));
        HIP_TRY(c.e_stuff_state.reserve(pd::fused_stuff_state_words(j.stream_cap) * 8));
        if (!c.h_totals) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c.h_totals), Context::kTotalsWords * 8, hipHostMallocDefault));
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
    if (!c.h_totals) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c.h_totals), Context::kTotalsWords * 8, hipHostMallocDefault));
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
    HIP_TRY(hipMemcpyAsync(counts, c.e_hist.p, pixo_host::kScanTableWords * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
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
            int rc = scan_count(c, j, stream, own);
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
                 const uint64_t *counts, bool wait = true)
{
    namespace pd = pixo_dev;
    {
        const int rc = scan_tables(c, j, o, g, stream, counts);
        if (rc) return rc;
    }
    if (j.fused) { // lengths, prefix and packing in one pass; the stream starts at bit 0 whatever the band's offset will be
        const bool zero = c.code_state_zero_words >= pd::fused_code_state_words(j.n);
        c.code_state_zero_words = 0; // (dirty from here until a stuffing launch has cleaned it)
        // chained with the stuffing kernel (!wait): this launch also zeroes that kernel's descriptors
        HIP_TRY(pd::launch_scan_code(j.a, c.e_code_state.as<unsigned long long>(), zero, c.e_stream.as<uint32_t>(),
                                     wait ? nullptr : c.e_stuff_state.as<unsigned long long>(),
                                     wait ? 0 : pd::fused_stuff_state_words(j.stream_cap), reinterpret_cast<unsigned long long *>(c.h_totals), stream));
        if (!wait) return PIXO_OK; // (the caller chains the stuffing kernel and synchronises once)
        HIP_TRY(hipStreamSynchronize(stream)); // (the kernel wrote the length into the pinned mailbox itself)
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

// The stuffing kernel of jpeg_scan_fused.hip over the packed stream (launch_scan_code has been enqueued; with
// `chained` its length has not been read back yet): afterwards c.e_out holds j.scan_bytes finished bytes.  The output
// buffer is sized from experience (grow-only) — the kernel never writes beyond it and says how much it needed.
// Where the stuffed bytes go when not into the context's device buffer: host memory the GPU can write (pinned), so that
// the kernel's stores ARE the transfer — no second pass over the file, no second synchronisation.
struct HostTarget {
    uint8_t *p = nullptr; // device-visible address of the first stuffed byte
    size_t cap = 0;       // bytes available from there
    bool grow = false;    // p lies in the context's own pinned file buffer: too small = reserve more and repeat
    size_t before = 0, after = 0; // (grow) bytes the file needs in front of / behind the stuffed bytes
};

int scan_stuff_fused(Context &c, ScanJob &j, hipStream_t stream, uint64_t band_bit_offset, uint32_t *head, int *tail_bits,
                     uint32_t *tail, bool chained = false, HostTarget *host = nullptr)
{
    namespace pd = pixo_dev;
    uint32_t shift = 0;
    if (j.band) {
        const uint64_t want = (8 - (band_bit_offset & 7)) & 7;
        j.head_bits = static_cast<int>(j.total_bits < want ? j.total_bits : want);
        shift = static_cast<uint32_t>(j.head_bits);
    }
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
        HIP_TRY(pd::launch_stuff_fused(c.e_stream.as<uint32_t>(), c.e_code_state.as<unsigned long long>(), pd::fused_code_state_words(j.n),
                                       shift, j.band, j.stream_cap, first_tile, tiles, c.e_stuff_state.as<unsigned long long>(),
                                       /*state_is_zero=*/
                                       chained && attempt == 0,
 out, out_cap, reinterpret_cast<unsigned long long *>(c.h_totals), stream));
        c.code_state_zero_words = pd::fused_code_state_words(j.n);
        // (no read-back copies: both kernels store their totals into the pinned mailbox h_totals — [0] bits of the scan,
        // [1] stuffed bytes, [2] packed bytes — which the host reads after the synchronisation below)
        uint32_t edge[3] = {0, 0, 0}; // band: stream word 0 (head bits) and the two words around the tail bits
        if (j.band) {
            const uint64_t tail_at = static_cast<uint64_t>(j.head_bits) + 8 * ((j.total_bits - j.head_bits) / 8);
            HIP_TRY(hipMemcpyAsync(&edge[0], c.e_stream.p, 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(&edge[1], c.e_stream.as<uint32_t>() + (tail_at >> 5), 8, hipMemcpyDeviceToHost, stream));
        }
        HIP_TRY(hipStreamSynchronize(stream));
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
int scan_pack(Context &c, ScanJob &j, hipStream_t stream, uint64_t band_bit_offset = 0, uint32_t *head = nullptr,
              int *tail_bits = nullptr, uint32_t *tail = nullptr)
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

// One uninterrupted scan of a large image, coded in PIECES so that the file's way to the host (0.21 of the 0.31 ms of a
// 4096x4096 noise image, 3.3 of 4.4 ms for 16384x16384) overlaps the coding: piece k = a run of consecutive groups,
// coded and stuffed by its own launch pair; the pieces hand each other the bit position and the byte position on the
// device (pixo_dev::ScanPiece), the host only waits for piece k's event to learn how many bytes it may copy — on a second
// stream — while piece k + 1 is being coded.  The bytes are the same as from one launch pair; a piece whose stream
// outgrows the guess its stuffing grid was sized for, or an output buffer that proves too small, sends the caller back
// to the one-piece path (return value 1; nothing of the result is kept).
bool pieces_enabled()
{ // PIXO_HIP_ONE_PIECE=1: always the one-piece path (A/B)
    static const bool on = [] { const char *e = std::getenv("PIXO_HIP_ONE_PIECE"); return !(e && *e && *e != '0'); }();
    return on;
}
constexpr uint32_t kMaxPieces = 16;
uint64_t piece_min_groups()
{ // a piece is at least this many groups of 192 blocks, and a scan of fewer than two such pieces is coded in one.
  // 2048 groups = the scan of a 4096x4096 4:2:0 image: the kernels of a smaller piece are mostly start-up — a sixth of
  // that scan takes 28 us where the whole takes 52 — and a 4096x4096 image in 2 or 6 pieces is no faster than in one
  // (0.30-0.33 against 0.31 ms); a 16384x16384 scan in 16 such pieces hides its 1 ms of coding behind 3.4 ms of PCIe.
  // (PIXO_HIP_PIECE_GROUPS=n: tests cut small images into many pieces to meet every alignment of the seams)
    static const uint64_t n = [] {
        const char *e = std::getenv("PIXO_HIP_PIECE_GROUPS");
        const long v = e ? std::atol(e) : 0;
        return static_cast<uint64_t>(v > 0 ? v : 2048);
    }();
    return n;
}

// Medium scans (piece_medium_groups() <= groups < 2 piece_min_groups()): relative sizes of the pieces, first to last
// (PIXO_HIP_PIECE_SCHEDULE="1,3"; "1" = one piece).  4096x4096 noise, 11 MB file, into pinned memory: one piece 0.313 ms,
// "1,3" 0.301, "1,2,5" 0.302, "1,2,3,4" 0.315; with the coefficient kernel band by band as well: "1,3" 0.291, "1,2,5" 0.299,
// "1,5" 0.310 (tools/gpu/r2w.sh): the first piece's bytes leave 60 us after the start instead of 100, the rest is the
// file's 0.21 ms on PCIe.
const std::vector<uint32_t> &piece_schedule()
{
    static const std::vector<uint32_t> w = [] {
        std::vector<uint32_t> v;
        const char *e = std::getenv("PIXO_HIP_PIECE_SCHEDULE");
        std::string t = e && *e ? e : "1,3";
        for (size_t i = 0; i < t.size();) {
            const size_t k = t.find(',', i);
            const long x = std::atol(t.substr(i, k == std::string::npos ? std::string::npos : k - i).c_str());
            if (x > 0) v.push_back(static_cast<uint32_t>(x));
            if (k == std::string::npos) break;
            i = k + 1;
        }
        if (v.empty()) v.push_back(1);
        return v;
    }();
    return w;
}
bool piece_medium_forced()
{ // (tests: PIXO_HIP_PIECE_MEDIUM set = medium scans in pieces whatever the last file's size)
    static const bool on = std::getenv("PIXO_HIP_PIECE_MEDIUM") != nullptr;
    return on;
}
uint64_t piece_medium_groups()
{ // scans of fewer groups than this are coded in one piece (PIXO_HIP_PIECE_MEDIUM=n)
    static const uint64_t n = [] {
        const char *e = std::getenv("PIXO_HIP_PIECE_MEDIUM");
        const long v = e ? std::atol(e) : 0;
        return static_cast<uint64_t>(v > 0 ? v : 1024);
    }();
    return n;
}

// Pixels whose coefficients have not been computed yet (the tuple's place is reserved, j.a points at it): the entropy
// stage launches the coefficient kernel itself — for a scan coded in pieces, band by band in front of each piece, so
// that the first piece's bytes can leave before the rest of the image has even been transformed.
struct PixelSource {
    const void *d_px;
    const pixo_jpeg_options *o;
    const pixo_host::Geometry *g;
    int16_t *dy, *dcb, *dcr;
};

int device_entropy_pieces(Context &c, ScanJob &j, hipStream_t stream, uint8_t *dst, size_t dst_cap, uint64_t *scan_bytes,
                          const PixelSource *src = nullptr)
{ // dst: where the stuffed scan goes on the host (dst_cap bytes available); tables are uploaded, j.a is set up
    namespace pd = pixo_dev;
    const uint64_t kGroupBlocks = 192, groups = (j.n + kGroupBlocks - 1) / kGroupBlocks;
    // where the pieces begin (in groups).  A large scan: equal pieces of at least piece_min_groups().  A medium one (a
    // 4096x4096 image): a few pieces that GROW — the first small, so that its bytes leave early, each next one coded
    // while the one before travels (weights from piece_schedule()).
    uint64_t begin[kMaxPieces + 1];
    uint32_t pieces = 0;
    if (groups >= 2 * piece_min_groups()) {
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
            const int rc = coeffs_rows(c, src->d_px, *src->o, *src->g, stream, src->dy, src->dcb, src->dcr, 0, 0);
            if (rc) return rc;
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
    for (uint32_t k = 0; k < pieces; ++k) {
        pd::ScanArgs a = j.a;
        a.nblocks = pc[k].blocks;
        a.pad_last = k + 1 == pieces ? 1u : 0u;
        const pd::ScanPiece piece{pc[k].first_block, k, bits_chain, k ? pc[k - 1].stream : nullptr};
        unsigned long long *mail = reinterpret_cast<unsigned long long *>(c.h_totals) + 4 * k;
        const bool zero = c.code_state_zero_words >= state_words;
        c.code_state_zero_words = 0;
        if (groups_per_row) { // this piece's MCU rows through the coefficient kernel
            const uint32_t row0 = static_cast<uint32_t>(begin[k] / groups_per_row);
            const uint32_t rows = k + 1 == pieces ? 0u : static_cast<uint32_t>(begin[k + 1] / groups_per_row) - row0;
            const int rc = coeffs_rows(c, src->d_px, *src->o, *src->g, stream, src->dy, src->dcb, src->dcr, row0, rows);
            if (rc) return rc;
        }
        HIP_TRY(pd::launch_scan_code(a, c.e_code_state.as<unsigned long long>(), zero, pc[k].stream, c.e_stuff_state.as<unsigned long long>(),
                                     pd::fused_stuff_state_words(j.stream_cap), mail, stream, &piece));
        HIP_TRY(pd::launch_stuff_fused(pc[k].stream, c.e_code_state.as<unsigned long long>(), state_words, 0, /*band=*/k + 1 != pieces,
                                       j.stream_cap, 0, pc[k].tiles, c.e_stuff_state.as<unsigned long long>(), /*state_is_zero=*/true,
                                       c.e_out.as<uint8_t>(), c.e_out.cap, mail, stream, out_chain, k));
        c.code_state_zero_words = state_words;
        HIP_TRY(hipEventRecord(c.piece_done[k], stream));
    }
    uint64_t done = 0; // bytes of the scan that are on their way to the host
    bool redo = false;
    sw.lap("  pieces enqueued");
    for (uint32_t k = 0; k < pieces; ++k) {
        HIP_TRY(hipEventSynchronize(c.piece_done[k]));
        sw.lap("  piece coded");
        if (redo) continue; // (still wait for everything that was enqueued)
        const uint64_t *mail = c.h_totals + 4 * k;
        const uint64_t stream_bits = mail[0], packed = k + 1 != pieces ? stream_bits / 8 : (stream_bits + 7) / 8;
        if (pd::stuff_tiles(packed) > pc[k].tiles) { redo = true; continue; } // (the stuffing grid was a guess: this piece is not complete)
        const uint64_t bytes = mail[1];
        if (done + bytes > c.e_out.cap || done + bytes > dst_cap) { redo = true; continue; }
        if (bytes) HIP_TRY(hipMemcpyAsync(dst + done, c.e_out.as<uint8_t>() + done, bytes, hipMemcpyDeviceToHost, c.copy_stream));
        done += bytes;
        sw.lap("  copy enqueued");
    }
    HIP_TRY(hipStreamSynchronize(c.copy_stream));
    sw.lap("  copies done");
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
int device_entropy_to_pinned(Context &c, const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
                             const pixo_host::Geometry &g, hipStream_t stream, const uint8_t **file, size_t *file_len,
                             uint32_t batch = 1, std::vector<uint64_t> *image_starts = nullptr, size_t *header_len = nullptr,
                             uint8_t *dest = nullptr, size_t dest_cap = 0, bool *own_malloc = nullptr, const PixelSource *src = nullptr)
{ // src != null: the tuple (dy, dcb, dcr = src's) has not been computed yet, see PixelSource.
  // dest != null: the file goes straight into the caller's storage (no pinned intermediate); when it does not
  // fit, *file_len says how much is needed and nothing is copied (PIXO_ERR_BUFFER_TOO_SMALL).
  // own_malloc != null (and no dest): the caller wants the file in malloc'd memory it will own — once the size is known
  // the block is allocated and the device-to-host copy goes straight into it (*own_malloc = true, *file = the block);
  // the copy into pageable memory runs at the link's rate, and what it saves is the second pass over the file from
  // the pinned buffer (tools/ubench/upload.cpp: 0.21 ms + a warm 11 MB memcpy, or 1.30 against 1.38 ms for new pages).
  // *own_malloc stays false when the file was assembled in the pinned buffer after all (a scan coded in pieces).
    if (own_malloc) *own_malloc = false;
    namespace pd = pixo_dev;
    Stopwatch sw;
    ScanJob j;
    int rc = scan_begin(c, j, dy, dcb, dcr, o, g, batch, nullptr);
    if (rc) return rc;
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
    // (Not for a caller that wants a malloc'd block of its own: the block would have to be allocated before the size is
    // known — 64 bytes per block, cut to size afterwards — and a block of a new size is new pages every call, which the
    // device-to-host copy has to fault in and pin: 20 ms instead of 0.7 for the 4096x4096 noise image.  One piece, the
    // exact size, recycled by malloc.)
    if (j.fused && batch == 1 && pieces_enabled() && !direct_host_stores() && (large || medium) && (!dest || dest_cap >= likely_most) &&
        !(own_malloc && !dest)) {
        if (src && o.optimize_huffman) { // (the statistics need the whole tuple)
            if ((rc = coeffs_rows(c, src->d_px, o, g, stream, src->dy, src->dcb, src->dcr, 0, 0))) return rc;
            src = nullptr;
        }
        if ((rc = scan_tables(c, j, o, g, stream, nullptr))) return rc;
        pixo_host::file_headers(head, o, j.h);
        const size_t hdr = head.size();
        uint8_t *buf = dest;
        size_t cap = dest_cap;
        if (!buf) {
            if ((rc = c.reserve_hfile(likely_most))) return rc;
            buf = c.h_file;
            cap = c.hfile_cap;
        }
        uint64_t scan_bytes = 0;
        rc = device_entropy_pieces(c, j, stream, buf + hdr, cap - hdr - 2, &scan_bytes, src);
        src = nullptr; // (the tuple is complete now, whatever happened)
        sw.lap("code+stuff+copy (pieces)");
        if (rc < 0) return rc;
        if (rc == 0) {
            c.packed_per_block = static_cast<uint32_t>(scan_bytes / (j.n ? j.n : 1));
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
    if (src && (rc = coeffs_rows(c, src->d_px, o, g, stream, src->dy, src->dcb, src->dcr, 0, 0))) return rc; // one piece: the whole image first
    if (j.fused) { // code + stuff back to back, one read-back
        if ((rc = scan_lengths(c, j, o, g, stream, nullptr, /*wait=*/false))) return rc;
        // One image into host memory the GPU can write — the context's pinned file buffer, or storage of the caller's
        // that is pinned / registered: the stuffing kernel stores straight into it, behind the place of the headers.
        HostTarget target;
        bool direct = false;
        if (batch == 1 && direct_host_stores()) {
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
        if ((rc = scan_stuff_fused(c, j, stream, 0, nullptr, nullptr, nullptr, /*chained=*/true, direct ? &target : nullptr))) return rc;
        sw.lap("code+stuff (fused)");
        if (direct) {
            const size_t hdr = head.size(), total = hdr + j.scan_bytes + 2;
            if (dest && j.scan_bytes > target.cap) {
                *file_len = total;
                return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(total) + " bytes");
            }
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
    if (batch == 1 && j.n) c.packed_per_block = static_cast<uint32_t>(scan_bytes / j.n);
    if (batch > 1) { // where every image's segment begins in the stuffed stream (reuses the seg_bytes buffer: 8 B/entry)
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
    uint8_t *buf = dest;
    bool mine = false;
    if (dest) {
        if (total > dest_cap) {
            *file_len = total;
            return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(total) + " bytes");
        }
    } else if (own_malloc && batch == 1) {
        buf = static_cast<uint8_t *>(std::malloc(total));
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

// Copies into FRESH host memory are page-fault bound (one core maps and fills a few GB/s of new pages):
// above a few MB the bytes are spread over a handful of threads (PIXO_HIP_COPY_THREADS, default 8; 1 = none).
unsigned copy_threads()
{
    static const unsigned n = [] {
        const char *e = std::getenv("PIXO_HIP_COPY_THREADS");
        const long v = e ? std::atol(e) : 8;
        return static_cast<unsigned>(v < 1 ? 1 : (v > 64 ? 64 : v));
    }();
    return n;
}
template <class F> void run_on_threads(unsigned t, F &&body) // body(index) for index in [0, t)
{
    if (t <= 1) { body(0u); return; }
    std::vector<std::thread> workers;
    workers.reserve(t - 1);
    for (unsigned i = 1; i < t; ++i) workers.emplace_back([&body, i] { body(i); });
    body(0u);
    for (auto &w : workers) w.join();
}
void big_copy(uint8_t *dst, const uint8_t *src, size_t n)
{
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
    uint8_t *p = static_cast<uint8_t *>(std::malloc(n ? n : 1));
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

// The seven scans of simple_progressive_script (progressive.rs:98-110) coded by the kernels of
// jpeg_entropy.hip over the device tuple; `out` already holds the file headers.  All scans are ONE
// packed stream of byte-aligned segments (the virtual block order is scan by scan, storage order inside
// a scan), so lengths / prefix sum / pack / 0xFF stuffing run once; the host only splices the seven SOS
// headers between the stuffed segments.
// The file is assembled in the context's pinned buffer: `head` (the file headers), then per scan its SOS header and its
// stuffed segment — every segment copied from the device straight to its final place — then EOI.  (Round 1 copied the
// stuffed stream to the host in one piece and spliced it into a std::vector, which the caller copied once more: two
// extra passes over the file through freshly mapped pages, about half of the 2.3 ms of a 4096x4096 preset-2 file.)
int device_progressive_scans(const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_host::Geometry &g,
                             const pixo_host::HuffSet &h, Context &c, const std::vector<uint8_t> &head, const uint8_t **file,
                             size_t *file_len)
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
    if (!c.h_totals) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c.h_totals), Context::kTotalsWords * 8, hipHostMallocDefault));
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
    int rc = c.reserve_hfile(total);
    if (rc) return rc;
    uint8_t *p = c.h_file;
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
                        std::vector<uint8_t> &spill, const uint8_t **file, size_t *file_len)
{
    namespace pd = pixo_dev;
    int rc;
    int16_t *dy = nullptr, *dcb = nullptr, *dcr = nullptr;
    const float *qt_all = nullptr;
    if ((rc = device_tables(c.device, &qt_all))) return rc;
    const float *qt = qt_all + (o.quality - 1) * pixo_host::kDeviceQtFloats;
    pixo_host::HuffSet h;
    const bool need_plain = o.optimize_huffman || !o.trellis_quant;
    if (need_plain && (rc = coeffs_on_device(c, d_pixels, o, g, c.stream, &dy, &dcb, &dcr))) return rc;
    if ((rc = huffman_for_tuple(dy, dcb, dcr, o, g, c, h))) return rc;
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
    if (!std::getenv("PIXO_HIP_HOST_ENTROPY")) {
        std::vector<uint8_t> head;
        pixo_host::file_headers(head, o, h);
        return device_progressive_scans(dy, dcb, dcr, g, h, c, head, file, file_len);
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

int hand_over(const std::vector<uint8_t> &v, uint8_t **out, size_t *out_len)
{
    uint8_t *p = static_cast<uint8_t *>(std::malloc(v.size() ? v.size() : 1));
    if (!p) return fail(PIXO_ERR_COMPRESSION, "Compression error: out of host memory");
    big_copy(p, v.data(), v.size());
    *out = p;
    *out_len = v.size();
    return PIXO_OK;
}

// Encodes host pixels; on return `*file` points at the finished file, either in the context's
// pinned buffer or in `spill` (host coder: scans with restart markers).
// dest / dest_cap / own_malloc: as for device_entropy_to_pinned (honoured by the baseline device path; the others
// return a view and the caller copies).
int encode_to_view(const uint8_t *data, size_t data_len, const pixo_jpeg_options &o, std::vector<uint8_t> &spill,
                   const uint8_t **file, size_t *file_len, uint8_t *dest = nullptr, size_t dest_cap = 0, bool *own_malloc = nullptr)
{
    if (own_malloc) *own_malloc = false;
    std::string msg;
    int rc = pixo_host::validate(o, true, data_len, msg);
    if (rc) return fail(rc, msg);
    if (!data) return fail(PIXO_ERR_COMPRESSION, "Compression error: null argument 'data'");
    const pixo_host::Geometry g = pixo_host::geometry(o.width, o.height, o.color_type, o.subsampling);
    Context &c = thread_context();
    if (!o.progressive && std::getenv("PIXO_HIP_HOST_ENTROPY")) { // (experiments: the host twin of the entropy stage)
        const int16_t *y, *cb, *cr;
        if ((rc = coeffs_to_pinned(c, data, o, g, &y, &cb, &cr))) return rc;
        pixo_host::encode_file(y, cb, cr, o, spill);
        *file = spill.data();
        *file_len = spill.size();
        return PIXO_OK;
    }
    if ((rc = c.ensure())) return rc;
    PIXO_ON_DEVICE_OF(c);
    const size_t px_bytes = static_cast<size_t>(o.width) * o.height * (g.gray ? 1 : 3);
    if ((rc = c.reserve_px((px_bytes + 15) & ~size_t{15}))) return rc;
    Stopwatch sw;
    HIP_TRY(hipMemcpyAsync(c.d_px, data, px_bytes, hipMemcpyHostToDevice, c.stream));
    sw.lap("pixels to device (enqueued)");
    if (o.progressive) return progressive_to_view(c.d_px, o, g, c, spill, file, file_len);
    int16_t *dy, *dcb, *dcr;
    if ((rc = coeffs_on_device(c, c.d_px, o, g, c.stream, &dy, &dcb, &dcr))) return rc;
    return device_entropy_to_pinned(c, dy, dcb, dcr, o, g, c.stream, file, file_len, 1, nullptr, nullptr, dest, dest_cap, own_malloc);
}

} // namespace

// A null pointer where the contract wants an object is a caller bug the Rust API cannot express; the C ABI
// answers it with an error instead of a crash.
namespace {
int fail_tuple_trellis()
{ // trellis quantisation happens between the transform and the tuple (src/jpeg/mod.rs:932-976): a tuple entry cannot apply it
    return fail(PIXO_ERR_COMPRESSION, "Compression error: trellis_quant needs the pixels: quantise the tuple with the trellis "
                                      "quantiser first and clear the flag, or use an entry point that takes pixels");
}
} // namespace
#define PIXO_REQUIRE(p) do { if (!(p)) return fail(PIXO_ERR_COMPRESSION, "Compression error: null argument '" #p "'"); } while (0)

extern "C" {

void pixo_jpeg_options_from_preset(pixo_jpeg_options *o, uint32_t width, uint32_t height,
                                   uint8_t quality, uint8_t preset)
{ // jpeg/mod.rs:162-216
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->width = width; o->height = height; o->color_type = PIXO_RGB; o->quality = quality;
    o->subsampling = PIXO_S444;
    if (preset == 0) return;
    o->optimize_huffman = 1;
    if (preset == 2) { o->subsampling = PIXO_S420; o->progressive = 1; o->trellis_quant = 1; }
}

int pixo_hip_jpeg_encode(const uint8_t *data, size_t data_len, const pixo_jpeg_options *options,
                         uint8_t **out, size_t *out_len)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(out);
    PIXO_REQUIRE(out_len);
    std::vector<uint8_t> spill;
    const uint8_t *file = nullptr;
    size_t n = 0;
    bool own = false;
    int rc = encode_to_view(data, data_len, *options, spill, &file, &n, nullptr, 0, &own);
    if (rc) return rc;
    if (own) { // (the device-to-host copy went straight into the block the caller gets)
        *out = const_cast<uint8_t *>(file);
        *out_len = n;
        return PIXO_OK;
    }
    Stopwatch sw;
    rc = deliver(file, n, out, out_len);
    sw.lap("file into fresh host memory");
    return rc;
}

int pixo_hip_jpeg_encode_into(uint8_t *output, size_t capacity, const uint8_t *data, size_t data_len,
                              const pixo_jpeg_options *options, size_t *out_len)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(out_len);
    if (capacity && !output) return fail(PIXO_ERR_COMPRESSION, "Compression error: null argument 'output'");
    std::vector<uint8_t> spill;
    const uint8_t *file = nullptr;
    size_t n = 0;
    static uint8_t nowhere; // (a null output with capacity 0 is a size query)
    int rc = encode_to_view(data, data_len, *options, spill, &file, &n, output ? output : &nowhere, output ? capacity : 0);
    if (rc == PIXO_OK || rc == PIXO_ERR_BUFFER_TOO_SMALL) *out_len = n; // (the size needed when the file does not fit)
    if (rc) return rc;
    if (file == output) return PIXO_OK; // (copied from the device straight into the caller's storage)
    if (n > capacity)
        return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(n) + " bytes");
    std::memcpy(output, file, n);
    return PIXO_OK;
}

int pixo_hip_encode_jpeg(const uint8_t *data, size_t data_len, uint32_t width, uint32_t height,
                         uint8_t color_type, uint8_t quality, uint8_t preset, int subsampling_420,
                         uint8_t **out, size_t *out_len)
{ // wasm.rs:113-142
    PIXO_REQUIRE(out);
    PIXO_REQUIRE(out_len);
    if (color_type != PIXO_GRAY && color_type != PIXO_RGB)
        return fail(PIXO_ERR_INVALID_COLOR_ARG, "Invalid color type for JPEG: " + std::to_string(color_type) +
                                                    ". Expected 0 (Gray) or 2 (Rgb)");
    pixo_jpeg_options o;
    pixo_jpeg_options_from_preset(&o, width, height, quality, preset); // .quality(q).preset(p)
    o.color_type = color_type;                                         // preset keeps the colour type
    o.subsampling = subsampling_420 ? PIXO_S420 : PIXO_S444;           // .subsampling(...) overrides
    return pixo_hip_jpeg_encode(data, data_len, &o, out, out_len);
}

int pixo_hip_coeff_geometry(uint32_t width, uint32_t height, uint8_t color_type, uint8_t subsampling,
                            size_t *y_blocks, size_t *c_blocks)
{
    PIXO_REQUIRE(y_blocks);
    PIXO_REQUIRE(c_blocks);
    if (width == 0 || height == 0)
        return fail(PIXO_ERR_INVALID_DIMENSIONS,
                    "Invalid image dimensions: " + std::to_string(width) + "x" + std::to_string(height));
    if (color_type != PIXO_GRAY && color_type != PIXO_RGB)
        return fail(PIXO_ERR_UNSUPPORTED_COLOR_TYPE, "Unsupported color type for this format");
    const pixo_host::Geometry g = pixo_host::geometry(width, height, color_type, subsampling);
    *y_blocks = g.y_blocks;
    *c_blocks = g.c_blocks;
    return PIXO_OK;
}

int pixo_hip_jpeg_coeffs(const uint8_t *pixels, uint32_t width, uint32_t height, uint8_t color_type,
                         uint8_t subsampling, uint8_t quality, int16_t *y, size_t y_blocks, int16_t *cb,
                         int16_t *cr, size_t c_blocks)
{
    pixo_jpeg_options o{};
    o.width = width; o.height = height; o.color_type = color_type; o.quality = quality;
    o.subsampling = subsampling;
    std::string msg;
    int rc = pixo_host::validate(o, false, 0, msg);
    if (rc) return fail(rc, msg);
    const pixo_host::Geometry g = pixo_host::geometry(width, height, color_type, subsampling);
    if (y_blocks != g.y_blocks || c_blocks != g.c_blocks)
        return fail(PIXO_ERR_INVALID_DATA_LENGTH,
                    "Invalid pixel data length: expected " + std::to_string(g.y_blocks) + " bytes, got " +
                        std::to_string(y_blocks));
    PIXO_REQUIRE(pixels);
    PIXO_REQUIRE(y);
    if (g.c_blocks && (!cb || !cr)) return fail(PIXO_ERR_COMPRESSION, "Compression error: null argument 'cb'/'cr'");
    const int16_t *hy, *hcb, *hcr;
    if ((rc = coeffs_to_pinned(thread_context(), pixels, o, g, &hy, &hcb, &hcr))) return rc;
    std::memcpy(y, hy, g.y_blocks * 128);
    if (g.c_blocks) {
        std::memcpy(cb, hcb, g.c_blocks * 128);
        std::memcpy(cr, hcr, g.c_blocks * 128);
    }
    return PIXO_OK;
}

int pixo_hip_jpeg_coeffs_device(const void *d_pixels, uint32_t width, uint32_t height, uint8_t color_type,
                                uint8_t subsampling, uint8_t quality, uint32_t batch, void *d_y, void *d_cb,
                                void *d_cr, void *stream)
{
    pixo_jpeg_options o{};
    o.width = width; o.height = height; o.color_type = color_type; o.quality = quality;
    o.subsampling = subsampling;
    std::string msg;
    int rc = pixo_host::validate(o, false, 0, msg);
    if (rc) return fail(rc, msg);
    if (batch == 0 || batch > 65535) return fail(PIXO_ERR_COMPRESSION, "Compression error: batch must be 1..65535");
    PIXO_REQUIRE(d_pixels);
    PIXO_REQUIRE(d_y);
    if (color_type != PIXO_GRAY && (!d_cb || !d_cr)) return fail(PIXO_ERR_COMPRESSION, "Compression error: null argument 'd_cb'/'d_cr'");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const float *qt_all = nullptr;
    if ((rc = device_tables(dev, &qt_all))) return rc;
    const bool gray = color_type == PIXO_GRAY;
    HIP_TRY(pixo_dev::launch_jpeg_coeffs(d_pixels, width, height, gray, !gray && subsampling == PIXO_S420,
                                         batch, d_y, gray ? nullptr : d_cb, gray ? nullptr : d_cr,
                                         qt_all + (quality - 1) * pixo_host::kDeviceQtFloats, static_cast<hipStream_t>(stream)));
    return PIXO_OK;
}

namespace {
int integer_mode_checks(uint32_t width, uint32_t height, uint8_t color_type, uint8_t subsampling, uint8_t quality,
                        pixo_host::QuantTables *qt)
{
    pixo_jpeg_options o{};
    o.width = width; o.height = height; o.color_type = color_type; o.quality = quality; o.subsampling = subsampling;
    std::string msg;
    int rc = pixo_host::validate(o, false, 0, msg);
    if (rc) return fail(rc, msg);
    if (color_type != PIXO_GRAY && subsampling != PIXO_S444)
        return fail(PIXO_ERR_COMPRESSION, "Compression error: the integer DCT mode is defined per 8x8 block: 4:4:4 or gray only");
    *qt = pixo_host::make_quant_tables(quality);
    return PIXO_OK;
}
} // namespace

int pixo_hip_jpeg_coeffs_integer_device(const void *d_pixels, uint32_t width, uint32_t height, uint8_t color_type,
                                        uint8_t subsampling, uint8_t quality, void *d_y, void *d_cb, void *d_cr, void *stream)
{
    pixo_host::QuantTables qt;
    int rc = integer_mode_checks(width, height, color_type, subsampling, quality, &qt);
    if (rc) return rc;
    PIXO_REQUIRE(d_pixels);
    PIXO_REQUIRE(d_y);
    const bool gray = color_type == PIXO_GRAY;
    if (!gray && (!d_cb || !d_cr)) return fail(PIXO_ERR_COMPRESSION, "Compression error: null argument 'd_cb'/'d_cr'");
    uint16_t ql[64], qc[64];
    for (int i = 0; i < 64; ++i) { ql[i] = static_cast<uint16_t>(qt.lum[i]); qc[i] = static_cast<uint16_t>(qt.chr[i]); } // quantize.rs:56-78
    HIP_TRY(pixo_dev::launch_jpeg_coeffs_integer(d_pixels, width, height, gray, ql, qc, d_y, d_cb, d_cr, static_cast<hipStream_t>(stream)));
    return PIXO_OK;
}

int pixo_hip_jpeg_coeffs_integer(const uint8_t *pixels, uint32_t width, uint32_t height, uint8_t color_type, uint8_t subsampling,
                                 uint8_t quality, int16_t *y, size_t y_blocks, int16_t *cb, int16_t *cr, size_t c_blocks)
{
    pixo_host::QuantTables qt;
    int rc = integer_mode_checks(width, height, color_type, subsampling, quality, &qt);
    if (rc) return rc;
    const pixo_host::Geometry g = pixo_host::geometry(width, height, color_type, PIXO_S444);
    if (y_blocks != g.y_blocks || c_blocks != g.c_blocks)
        return fail(PIXO_ERR_INVALID_DATA_LENGTH, "Invalid pixel data length: expected " + std::to_string(g.y_blocks) + " bytes, got " +
                                                      std::to_string(y_blocks));
    PIXO_REQUIRE(pixels);
    PIXO_REQUIRE(y);
    if (g.c_blocks && (!cb || !cr)) return fail(PIXO_ERR_COMPRESSION, "Compression error: null argument 'cb'/'cr'");
    Context &c = thread_context();
    if ((rc = c.ensure())) return rc;
    PIXO_ON_DEVICE_OF(c);
    const size_t px_bytes = static_cast<size_t>(width) * height * (g.gray ? 1 : 3), coef_bytes = (g.y_blocks + 2 * g.c_blocks) * 128;
    if ((rc = c.reserve_px((px_bytes + 15) & ~size_t{15}))) return rc;
    if ((rc = c.reserve_coef(coef_bytes))) return rc;
    if ((rc = c.reserve_hcoef(coef_bytes))) return rc;
    HIP_TRY(hipMemcpyAsync(c.d_px, pixels, px_bytes, hipMemcpyHostToDevice, c.stream));
    int16_t *dy = static_cast<int16_t *>(c.d_coef), *dcb = dy + g.y_blocks * 64, *dcr = dcb + g.c_blocks * 64;
    if ((rc = pixo_hip_jpeg_coeffs_integer_device(c.d_px, width, height, color_type, PIXO_S444, quality, dy, dcb, dcr, c.stream))) return rc;
    HIP_TRY(hipMemcpyAsync(c.h_coef, c.d_coef, coef_bytes, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    const int16_t *hy = static_cast<const int16_t *>(c.h_coef);
    std::memcpy(y, hy, g.y_blocks * 128);
    if (g.c_blocks) {
        std::memcpy(cb, hy + g.y_blocks * 64, g.c_blocks * 128);
        std::memcpy(cr, hy + (g.y_blocks + g.c_blocks) * 64, g.c_blocks * 128);
    }
    return PIXO_OK;
}

int pixo_hip_jpeg_entropy_encode(const int16_t *y, const int16_t *cb, const int16_t *cr,
                                 const pixo_jpeg_options *options, uint8_t **out, size_t *out_len)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(out);
    PIXO_REQUIRE(out_len);
    std::string msg;
    int rc = pixo_host::validate(*options, false, 0, msg);
    if (rc) return fail(rc, msg);
    if (options->progressive && options->trellis_quant) return fail_tuple_trellis();
    PIXO_REQUIRE(y);
    std::vector<uint8_t> v;
    pixo_host::encode_file(y, cb, cr, *options, v);
    return hand_over(v, out, out_len);
}

namespace {
// Device-pointer entry points run on the context's own stream.  What the caller enqueued before the call
// — on the stream it named with pixo_hip_set_producer_stream, by default the NULL stream — is ordered
// in front of it with an event (no host synchronisation).
thread_local hipStream_t t_producer = nullptr;
int order_after_producer(Context &c)
{
    if (!c.producer_done) HIP_TRY(hipEventCreateWithFlags(&c.producer_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c.producer_done, t_producer));
    HIP_TRY(hipStreamWaitEvent(c.stream, c.producer_done, 0));
    return PIXO_OK;
}

// binds the thread's context to the HIP device that is current for the caller
int context_on_current_device(Context **out)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (t_slot.device != dev) (void)pixo_hip_set_device(dev);
    Context &c = thread_context();
    int rc = c.ensure();
    if (rc) return rc;
    if ((rc = order_after_producer(c))) return rc;
    *out = &c;
    return PIXO_OK;
}

int device_tuple_to_malloc(const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
                           const pixo_host::Geometry &g, Context &c, uint8_t **out, size_t *out_len)
{
    if (!std::getenv("PIXO_HIP_HOST_ENTROPY")) {
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
} // namespace

int pixo_hip_jpeg_entropy_encode_device(const void *d_y, const void *d_cb, const void *d_cr,
                                        const pixo_jpeg_options *options, uint8_t **out, size_t *out_len)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(out);
    PIXO_REQUIRE(out_len);
    std::string msg;
    int rc = pixo_host::validate(*options, false, 0, msg);
    if (rc) return fail(rc, msg);
    if (options->progressive && options->trellis_quant) return fail_tuple_trellis();
    PIXO_REQUIRE(d_y);
    Context *c = nullptr;
    if ((rc = context_on_current_device(&c))) return rc;
    const pixo_host::Geometry g = pixo_host::geometry(options->width, options->height, options->color_type, options->subsampling);
    return device_tuple_to_malloc(static_cast<const int16_t *>(d_y), static_cast<const int16_t *>(d_cb),
                                  static_cast<const int16_t *>(d_cr), *options, g, *c, out, out_len);
}

int pixo_hip_jpeg_encode_device(const void *d_pixels, const pixo_jpeg_options *options, uint8_t **out, size_t *out_len)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(d_pixels);
    PIXO_REQUIRE(out);
    PIXO_REQUIRE(out_len);
    std::string msg;
    int rc = pixo_host::validate(*options, false, 0, msg);
    if (rc) return fail(rc, msg);
    Context *c = nullptr;
    if ((rc = context_on_current_device(&c))) return rc;
    const pixo_host::Geometry g = pixo_host::geometry(options->width, options->height, options->color_type, options->subsampling);
    if (options->progressive) {
        std::vector<uint8_t> spill;
        const uint8_t *file = nullptr;
        size_t n = 0;
        if ((rc = progressive_to_view(d_pixels, *options, g, *c, spill, &file, &n))) return rc;
        return deliver(file, n, out, out_len);
    }
    int16_t *dy, *dcb, *dcr;
    if ((rc = coeffs_on_device(*c, d_pixels, *options, g, c->stream, &dy, &dcb, &dcr))) return rc;
    return device_tuple_to_malloc(dy, dcb, dcr, *options, g, *c, out, out_len);
}

int pixo_hip_jpeg_encode_device_into(const void *d_pixels, const pixo_jpeg_options *options, uint8_t *output, size_t capacity,
                                     size_t *out_len)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(out_len);
    std::string msg;
    int rc = pixo_host::validate(*options, false, 0, msg);
    if (rc) return fail(rc, msg);
    Context *c = nullptr;
    if ((rc = context_on_current_device(&c))) return rc;
    const pixo_host::Geometry g = pixo_host::geometry(options->width, options->height, options->color_type, options->subsampling);
    if (options->progressive) { // assembled in the context's pinned buffer: one copy from there if it fits
        PIXO_REQUIRE(d_pixels);
        std::vector<uint8_t> spill;
        const uint8_t *file = nullptr;
        size_t n = 0;
        if ((rc = progressive_to_view(d_pixels, *options, g, *c, spill, &file, &n))) return rc;
        *out_len = n;
        if (n > capacity) return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(n) + " bytes");
        std::memcpy(output, file, n);
        return PIXO_OK;
    }
    if (std::getenv("PIXO_HIP_HOST_ENTROPY")) { // assembled on the host: copy if it fits
        uint8_t *p = nullptr;
        size_t n = 0;
        if ((rc = pixo_hip_jpeg_encode_device(d_pixels, options, &p, &n))) return rc;
        *out_len = n;
        if (n > capacity) {
            std::free(p);
            return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(n) + " bytes");
        }
        std::memcpy(output, p, n);
        std::free(p);
        return PIXO_OK;
    }
    PIXO_REQUIRE(d_pixels);
    int16_t *dy, *dcb, *dcr;
    if ((rc = coeffs_reserve(*c, g, &dy, &dcb, &dcr))) return rc;
    const PixelSource src{d_pixels, options, &g, dy, dcb, dcr}; // (the entropy stage launches the coefficient kernel: whole, or band by band)
    const uint8_t *file = nullptr;
    // (a null output with capacity 0 is a size query)
    static uint8_t nowhere;
    return device_entropy_to_pinned(*c, dy, dcb, dcr, *options, g, c->stream, &file, out_len, 1, nullptr, nullptr,
                                    output ? output : &nowhere, output ? capacity : 0, nullptr, &src);
}

namespace {
// Argument checks shared by the two PNG entries; resolves the strategy the reference would run.
int png_plan(uint32_t width, uint32_t height, uint32_t bpp, uint8_t strategy, uint32_t flags, int *run, bool *sequential_fast)
{
    if (width == 0 || height == 0)
        return fail(PIXO_ERR_INVALID_DIMENSIONS, "Invalid image dimensions: " + std::to_string(width) + "x" + std::to_string(height));
    if (!(bpp == 1 || bpp == 2 || bpp == 3 || bpp == 4 || bpp == 6 || bpp == 8))
        return fail(PIXO_ERR_UNSUPPORTED_COLOR_TYPE, "Unsupported color type for this format");
    if (strategy > PIXO_PNG_BIGRAMS) return fail(PIXO_ERR_COMPRESSION, "Compression error: unknown PNG filter strategy");
    int s = strategy;
    const uint64_t area = static_cast<uint64_t>(width) * height;
    const bool adaptive = s == PIXO_PNG_ADAPTIVE || s == PIXO_PNG_ADAPTIVE_FAST || s == PIXO_PNG_BIGRAMS;
    if (area <= 4096 && adaptive) s = PIXO_PNG_SUB; // src/png/filter.rs:76-86
    // the stateful AdaptiveFast runs wherever the reference does not take its rayon path (:94-112)
    *sequential_fast = s == PIXO_PNG_ADAPTIVE_FAST && ((flags & PIXO_PNG_NO_RAYON) || height <= 32);
    *run = s;
    return PIXO_OK;
}

// zlib Adler-32 of the filtered stream from the per-row sums (A = byte sum, B = sum of
// (row_len - i) * byte_i): s2 += row_len * s1 + B, s1 += A  (mod 65521)
uint32_t combine_adler(const unsigned long long *sums, uint32_t height, uint64_t out_row_bytes)
{
    const uint64_t M = 65521;
    uint64_t s1 = 1, s2 = 0;
    const uint64_t L = out_row_bytes % M;
    for (uint32_t y = 0; y < height; ++y) {
        s2 = (s2 + L * s1 + sums[2 * y + 1] % M) % M;
        s1 = (s1 + sums[2 * y] % M) % M;
    }
    return static_cast<uint32_t>((s2 << 16) | s1);
}

int png_filter_on_device(Context &c, const void *d_in, uint32_t width, uint32_t height, uint32_t bpp, int run,
                         bool sequential_fast, void *d_out, uint32_t *adler)
{
    HIP_TRY(c.p_sums.reserve(static_cast<size_t>(height) * 16));
    HIP_TRY(c.p_scratch.reserve(16));
    if (static_cast<size_t>(height) * 16 > c.hsums_cap) {
        if (c.h_sums) (void)hipHostFree(c.h_sums);
        c.h_sums = nullptr; c.hsums_cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c.h_sums), static_cast<size_t>(height) * 16, hipHostMallocDefault));
        c.hsums_cap = static_cast<size_t>(height) * 16;
    }
    HIP_TRY(pixo_dev::launch_png_filter(d_in, width, height, bpp, run, sequential_fast, d_out,
                                        c.p_sums.as<unsigned long long>(), c.p_scratch.as<int>(), c.stream));
    HIP_TRY(hipMemcpyAsync(c.h_sums, c.p_sums.p, static_cast<size_t>(height) * 16, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    *adler = combine_adler(c.h_sums, height, static_cast<uint64_t>(width) * bpp + 1);
    return PIXO_OK;
}
} // namespace

int pixo_hip_png_filter(const uint8_t *data, size_t data_len, uint32_t width, uint32_t height, uint32_t bytes_per_pixel,
                        uint8_t strategy, uint32_t flags, uint8_t *out, size_t out_capacity, uint32_t *adler32)
{
    int run = 0;
    bool seq = false;
    int rc = png_plan(width, height, bytes_per_pixel, strategy, flags, &run, &seq);
    if (rc) return rc;
    const size_t in_bytes = static_cast<size_t>(width) * height * bytes_per_pixel;
    const size_t out_bytes = static_cast<size_t>(height) * (static_cast<size_t>(width) * bytes_per_pixel + 1);
    if (data_len != in_bytes)
        return fail(PIXO_ERR_INVALID_DATA_LENGTH, "Invalid pixel data length: expected " + std::to_string(in_bytes) +
                                                      " bytes, got " + std::to_string(data_len));
    if (out_capacity < out_bytes)
        return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(out_bytes) + " bytes");
    if (!data || !out || !adler32) return fail(PIXO_ERR_COMPRESSION, "Compression error: null argument");
    Context &c = thread_context();
    if ((rc = c.ensure())) return rc;
    PIXO_ON_DEVICE_OF(c);
    HIP_TRY(c.p_in.reserve((in_bytes + 15) & ~size_t{15}));
    HIP_TRY(c.p_out.reserve(out_bytes));
    HIP_TRY(hipMemcpyAsync(c.p_in.p, data, in_bytes, hipMemcpyHostToDevice, c.stream));
    if ((rc = png_filter_on_device(c, c.p_in.p, width, height, bytes_per_pixel, run, seq, c.p_out.p, adler32))) return rc;
    HIP_TRY(hipMemcpy(out, c.p_out.p, out_bytes, hipMemcpyDeviceToHost));
    return PIXO_OK;
}

int pixo_hip_png_filter_async(const void *d_data, uint32_t width, uint32_t height, uint32_t bytes_per_pixel,
                              uint8_t strategy, uint32_t flags, void *d_out, void *d_row_sums, void *d_scratch,
                              void *stream)
{
    int run = 0;
    bool seq = false;
    int rc = png_plan(width, height, bytes_per_pixel, strategy, flags, &run, &seq);
    if (rc) return rc;
    HIP_TRY(pixo_dev::launch_png_filter(d_data, width, height, bytes_per_pixel, run, seq, d_out,
                                        static_cast<unsigned long long *>(d_row_sums), static_cast<int *>(d_scratch),
                                        static_cast<hipStream_t>(stream)));
    return PIXO_OK;
}

uint32_t pixo_hip_png_adler32_from_row_sums(const uint64_t *row_sums, uint32_t width, uint32_t height,
                                            uint32_t bytes_per_pixel)
{
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "u64");
    return combine_adler(reinterpret_cast<const unsigned long long *>(row_sums), height,
                         static_cast<uint64_t>(width) * bytes_per_pixel + 1);
}

int pixo_hip_png_filter_device(const void *d_data, uint32_t width, uint32_t height, uint32_t bytes_per_pixel,
                               uint8_t strategy, uint32_t flags, void *d_out, uint32_t *adler32)
{
    int run = 0;
    bool seq = false;
    int rc = png_plan(width, height, bytes_per_pixel, strategy, flags, &run, &seq);
    if (rc) return rc;
    PIXO_REQUIRE(d_data);
    PIXO_REQUIRE(d_out);
    PIXO_REQUIRE(adler32);
    Context *c = nullptr;
    if ((rc = context_on_current_device(&c))) return rc;
    return png_filter_on_device(*c, d_data, width, height, bytes_per_pixel, run, seq, d_out, adler32);
}

int pixo_hip_jpeg_encode_batch_device(const void *d_pixels, const pixo_jpeg_options *options, uint32_t batch,
                                      uint8_t **files, size_t *lens)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(files);
    PIXO_REQUIRE(lens);
    std::string msg;
    int rc = pixo_host::validate(*options, false, 0, msg);
    if (rc) return fail(rc, msg);
    if (batch == 0 || batch > 65535) return fail(PIXO_ERR_COMPRESSION, "Compression error: batch must be 1..65535");
    Context *c = nullptr;
    if ((rc = context_on_current_device(&c))) return rc;
    const pixo_jpeg_options &o = *options;
    const pixo_host::Geometry g = pixo_host::geometry(o.width, o.height, o.color_type, o.subsampling);
    const size_t px_bytes = static_cast<size_t>(o.width) * o.height * (g.gray ? 1 : 3);
    for (uint32_t i = 0; i < batch; ++i) { files[i] = nullptr; lens[i] = 0; }
    auto release = [&](int code) { for (uint32_t i = 0; i < batch; ++i) { std::free(files[i]); files[i] = nullptr; } return code; };
    // Per-image tables or restart segments inside the images: one image at a time.
    if (o.progressive) {
        for (uint32_t i = 0; i < batch; ++i)
            if ((rc = pixo_hip_jpeg_encode_device(static_cast<const uint8_t *>(d_pixels) + i * px_bytes, options, &files[i], &lens[i]))) return release(rc);
        return PIXO_OK;
    }
    if (batch == 1 || o.optimize_huffman || scan_has_restart_markers(o, g) || px_bytes % 4 != 0) {
        for (uint32_t i = 0; i < batch; ++i) {
            int16_t *dy, *dcb, *dcr;
            if ((rc = coeffs_on_device(*c, static_cast<const uint8_t *>(d_pixels) + i * px_bytes, o, g, c->stream, &dy, &dcb, &dcr))) return release(rc);
            if ((rc = device_tuple_to_malloc(dy, dcb, dcr, o, g, *c, &files[i], &lens[i]))) return release(rc);
        }
        return PIXO_OK;
    }
    // one coefficient launch and one entropy pass for the whole batch
    const float *qt_all = nullptr;
    if ((rc = device_tables(c->device, &qt_all))) return rc;
    const size_t coef_bytes = (g.y_blocks + 2 * g.c_blocks) * 128 * batch;
    if ((rc = c->reserve_coef(coef_bytes))) return rc;
    int16_t *dy = static_cast<int16_t *>(c->d_coef), *dcb = dy + g.y_blocks * 64 * batch, *dcr = dcb + g.c_blocks * 64 * batch;
    HIP_TRY(pixo_dev::launch_jpeg_coeffs(d_pixels, o.width, o.height, g.gray, g.s420, batch, dy, g.gray ? nullptr : dcb,
                                         g.gray ? nullptr : dcr, qt_all + (o.quality - 1) * pixo_host::kDeviceQtFloats, c->stream));
    const uint8_t *blob = nullptr;
    size_t blob_len = 0, hdr = 0;
    std::vector<uint64_t> starts;
    if ((rc = device_entropy_to_pinned(*c, dy, dcb, dcr, o, g, c->stream, &blob, &blob_len, batch, &starts, &hdr))) return rc;
    for (uint32_t i = 0; i < batch; ++i) {
        lens[i] = hdr + static_cast<size_t>(starts[i + 1] - starts[i]) + 2;
        files[i] = static_cast<uint8_t *>(std::malloc(lens[i]));
        if (!files[i]) return release(fail(PIXO_ERR_COMPRESSION, "Compression error: out of host memory"));
    }
    // headers + own segment + EOI into every file; the files are fresh memory: several threads (see big_copy)
    const size_t total = blob_len + static_cast<size_t>(batch) * hdr;
    const unsigned t = static_cast<unsigned>(std::min<size_t>(std::min<size_t>(copy_threads(), batch), total >> 21));
    run_on_threads(t ? t : 1, [&](unsigned k) {
        for (uint32_t i = k; i < batch; i += (t ? t : 1)) {
            const size_t seg = lens[i] - hdr - 2;
            uint8_t *p = files[i];
            std::memcpy(p, blob, hdr);
            std::memcpy(p + hdr, blob + hdr + starts[i], seg);
            p[hdr + seg] = 0xFF; p[hdr + seg + 1] = 0xD9;
        }
    });
    return PIXO_OK;
}

// ======================================================================================================
// One image across several GPUs: per-band entropy coding + splice
// ======================================================================================================
struct pixo_hip_band_encoder {
    pixo_jpeg_options image{}, band{};   // the whole image / the same options with the band's height
    pixo_host::Geometry g{};             // of the band
    uint32_t parts = 1, index = 0, rows = 0, row_begin = 0;
    Context *c = nullptr;                // adopted from the pool for the encoder's lifetime
    int16_t *dy = nullptr, *dcb = nullptr, *dcr = nullptr;
    ScanJob job;
    int stage = 0;                       // 0 created, 1 coefficients done, 2 lengths done
    int16_t last_dc[3] = {0, 0, 0};
};

namespace {
int band_rows(const pixo_jpeg_options &o, uint32_t parts, uint32_t index, uint32_t *row_begin, uint32_t *row_end)
{
    size_t yo, yb, co, cb;
    return pixo_hip_band(o.width, o.height, o.color_type, o.subsampling, parts, index, row_begin, row_end, &yo, &yb, &co, &cb);
}
bool band_codable(const pixo_jpeg_options &o, const pixo_host::Geometry &g)
{ // what a band encoder can do on its own: one uninterrupted baseline scan
    return !o.progressive && !scan_has_restart_markers(o, g);
}
} // namespace

int pixo_hip_band_encoder_create(const pixo_jpeg_options *options, uint32_t parts, uint32_t index, int device,
                                 pixo_hip_band_encoder **out)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(out);
    *out = nullptr;
    std::string msg;
    int rc = pixo_host::validate(*options, false, 0, msg);
    if (rc) return fail(rc, msg);
    if (parts == 0 || index >= parts) return fail(PIXO_ERR_COMPRESSION, "Compression error: bad band index");
    const pixo_host::Geometry whole = pixo_host::geometry(options->width, options->height, options->color_type, options->subsampling);
    if (!band_codable(*options, whole))
        return fail(PIXO_ERR_COMPRESSION, "Compression error: bands are entropy-coded on their own only for baseline scans without "
                                          "restart markers (gather the coefficient bands and use pixo_hip_jpeg_entropy_encode_device)");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n)
        return fail(PIXO_ERR_COMPRESSION, "Compression error: no HIP device " + std::to_string(device));
    std::unique_ptr<pixo_hip_band_encoder> e(new pixo_hip_band_encoder);
    e->image = *options; e->parts = parts; e->index = index;
    uint32_t r0 = 0, r1 = 0;
    if ((rc = band_rows(*options, parts, index, &r0, &r1))) return rc;
    e->row_begin = r0; e->rows = r1 - r0;
    e->band = *options;
    e->band.height = e->rows ? e->rows : 1;
    e->band.has_restart_interval = 0; e->band.restart_interval = 0;
    e->g = pixo_host::geometry(e->band.width, e->band.height, e->band.color_type, e->band.subsampling);
    if (e->rows == 0) { e->g.y_blocks = e->g.c_blocks = e->g.units = 0; e->g.units_y = 0; }
    e->c = pool().take(device);
    if ((rc = e->c->ensure())) { pool().give(e->c); return rc; }
    *out = e.release();
    return PIXO_OK;
}

void pixo_hip_band_encoder_destroy(pixo_hip_band_encoder *e)
{
    if (!e) return;
    if (e->c) {
        if (e->c->ready && e->c->stream) { DeviceScope on(e->c->device); (void)hipStreamSynchronize(e->c->stream); }
        pool().give(e->c);
    }
    delete e;
}

int pixo_hip_band_encoder_rows(const pixo_hip_band_encoder *e, uint32_t *row_begin, uint32_t *row_end)
{
    PIXO_REQUIRE(e);
    PIXO_REQUIRE(row_begin);
    PIXO_REQUIRE(row_end);
    *row_begin = e->row_begin; *row_end = e->row_begin + e->rows;
    return PIXO_OK;
}

int pixo_hip_band_encoder_coeffs(pixo_hip_band_encoder *e, const void *band_pixels, int on_device, int16_t last_dc[3])
{
    PIXO_REQUIRE(e);
    PIXO_REQUIRE(last_dc);
    Context &c = *e->c;
    PIXO_ON_DEVICE_OF(c);
    e->stage = 1;
    last_dc[0] = last_dc[1] = last_dc[2] = 0;
    if (e->rows == 0) return PIXO_OK; // more bands than MCU rows: nothing to do, the caller forwards the DCs above
    PIXO_REQUIRE(band_pixels);
    int rc;
    const void *d_px = band_pixels;
    if (!on_device) { // the band's rows come over this GPU's own PCIe link
        const size_t px_bytes = static_cast<size_t>(e->band.width) * e->rows * (e->g.gray ? 1 : 3);
        if ((rc = c.reserve_px((px_bytes + 15) & ~size_t{15}))) return rc;
        HIP_TRY(hipMemcpyAsync(c.d_px, band_pixels, px_bytes, hipMemcpyHostToDevice, c.stream));
        d_px = c.d_px;
    } else if ((rc = order_after_producer(c))) {
        return rc;
    }
    if ((rc = coeffs_on_device(c, d_px, e->band, e->g, c.stream, &e->dy, &e->dcb, &e->dcr))) return rc;
    // the DCs the next band predicts from: first coefficient of the last block of every plane
    if (!c.h_totals) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c.h_totals), Context::kTotalsWords * 8, hipHostMallocDefault));
    int16_t *h = reinterpret_cast<int16_t *>(c.h_totals);
    HIP_TRY(hipMemcpyAsync(h, e->dy + (e->g.y_blocks - 1) * 64, 2, hipMemcpyDeviceToHost, c.stream));
    if (e->g.c_blocks) {
        HIP_TRY(hipMemcpyAsync(h + 1, e->dcb + (e->g.c_blocks - 1) * 64, 2, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipMemcpyAsync(h + 2, e->dcr + (e->g.c_blocks - 1) * 64, 2, hipMemcpyDeviceToHost, c.stream));
    }
    HIP_TRY(hipStreamSynchronize(c.stream));
    e->last_dc[0] = h[0];
    e->last_dc[1] = e->g.c_blocks ? h[1] : 0;
    e->last_dc[2] = e->g.c_blocks ? h[2] : 0;
    for (int i = 0; i < 3; ++i) last_dc[i] = e->last_dc[i];
    return PIXO_OK;
}

int pixo_hip_band_encoder_count(pixo_hip_band_encoder *e, const int16_t prev_dc[3], uint64_t counts[PIXO_HIP_COUNT_WORDS])
{
    PIXO_REQUIRE(e);
    PIXO_REQUIRE(prev_dc);
    PIXO_REQUIRE(counts);
    if (e->stage < 1) return fail(PIXO_ERR_COMPRESSION, "Compression error: band encoder: coefficients first");
    Context &c = *e->c;
    PIXO_ON_DEVICE_OF(c);
    std::memset(counts, 0, sizeof(uint64_t) * PIXO_HIP_COUNT_WORDS);
    if (e->rows == 0) return PIXO_OK;
    int rc = scan_begin(c, e->job, e->dy, e->dcb, e->dcr, e->band, e->g, 1, prev_dc);
    if (rc) return rc;
    return scan_count(c, e->job, c.stream, counts);
}

int pixo_hip_band_encoder_lengths(pixo_hip_band_encoder *e, const int16_t prev_dc[3], const uint64_t *total_counts, uint64_t *bits)
{
    PIXO_REQUIRE(e);
    PIXO_REQUIRE(prev_dc);
    PIXO_REQUIRE(bits);
    if (e->stage < 1) return fail(PIXO_ERR_COMPRESSION, "Compression error: band encoder: coefficients first");
    if (e->image.optimize_huffman && !total_counts)
        return fail(PIXO_ERR_COMPRESSION, "Compression error: band encoder: optimised tables need the statistics of all bands");
    Context &c = *e->c;
    PIXO_ON_DEVICE_OF(c);
    int rc = scan_begin(c, e->job, e->dy, e->dcb, e->dcr, e->band, e->g, 1, prev_dc);
    if (rc) return rc;
    if ((rc = scan_lengths(c, e->job, e->image, e->g, c.stream, total_counts))) return rc;
    *bits = e->job.total_bits;
    e->stage = 2;
    return PIXO_OK;
}

int pixo_hip_band_encoder_pack_device(pixo_hip_band_encoder *e, uint64_t bit_offset, uint8_t header[16], void **d_body,
                                      size_t *body_len)
{
    PIXO_REQUIRE(e);
    PIXO_REQUIRE(header);
    if (e->stage < 2) return fail(PIXO_ERR_COMPRESSION, "Compression error: band encoder: lengths first");
    Context &c = *e->c;
    PIXO_ON_DEVICE_OF(c);
    uint32_t head = 0, tail = 0;
    int tail_bits = 0;
    int rc = scan_pack(c, e->job, c.stream, bit_offset, &head, &tail_bits, &tail);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c.stream)); // the body is complete in the encoder's device buffer
    std::vector<uint8_t> hdr;
    pixo_host::make_piece(hdr, e->job.head_bits, head, tail_bits, tail, nullptr, 0);
    const uint64_t body = e->job.scan_bytes;
    for (int i = 0; i < 8; ++i) hdr[8 + i] = static_cast<uint8_t>(body >> (8 * i));
    std::memcpy(header, hdr.data(), pixo_host::kPieceHeader);
    if (d_body) *d_body = c.e_out.p;
    if (body_len) *body_len = static_cast<size_t>(body);
    e->stage = 3;
    return PIXO_OK;
}

int pixo_hip_band_encoder_copy_body(pixo_hip_band_encoder *e, uint8_t *dst)
{
    PIXO_REQUIRE(e);
    if (e->stage < 3) return fail(PIXO_ERR_COMPRESSION, "Compression error: band encoder: pack first");
    Context &c = *e->c;
    PIXO_ON_DEVICE_OF(c);
    const size_t body = static_cast<size_t>(e->job.scan_bytes);
    if (!body) return PIXO_OK;
    PIXO_REQUIRE(dst);
    hipPointerAttribute_t attr;
    const bool known = hipPointerGetAttributes(&attr, dst) == hipSuccess;
    if (!known) (void)hipGetLastError(); // (plain malloc memory is "invalid value" to the runtime)
    if (known && (attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeDevice)) {
        // registered / hipHostMalloc'd storage: the device-to-host copy is the only pass over the bytes;
        // device storage (e.g. the send buffer of a collective): device to device
        HIP_TRY(hipMemcpyAsync(dst, c.e_out.p, body, hipMemcpyDefault, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
        return PIXO_OK;
    }
    // pageable destination: through the context's pinned buffer (a direct copy makes the runtime pin the pages first)
    int rc = c.reserve_hfile(body);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c.h_file, c.e_out.p, body, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    std::memcpy(dst, c.h_file, body);
    return PIXO_OK;
}

int pixo_hip_band_encoder_pack(pixo_hip_band_encoder *e, uint64_t bit_offset, uint8_t **piece, size_t *piece_len)
{
    PIXO_REQUIRE(piece);
    PIXO_REQUIRE(piece_len);
    uint8_t header[pixo_host::kPieceHeader];
    size_t body = 0;
    int rc = pixo_hip_band_encoder_pack_device(e, bit_offset, header, nullptr, &body);
    if (rc) return rc;
    uint8_t *p = static_cast<uint8_t *>(std::malloc(pixo_host::kPieceHeader + body));
    if (!p) return fail(PIXO_ERR_COMPRESSION, "Compression error: out of host memory");
    std::memcpy(p, header, pixo_host::kPieceHeader);
    if ((rc = pixo_hip_band_encoder_copy_body(e, p + pixo_host::kPieceHeader))) { std::free(p); return rc; }
    *piece = p;
    *piece_len = pixo_host::kPieceHeader + body;
    return PIXO_OK;
}

namespace {
int tables_for_splice(const pixo_jpeg_options &o, const uint64_t *total_counts, pixo_host::HuffSet &h)
{
    h = pixo_host::HuffSet::standard();
    if (!o.optimize_huffman) return PIXO_OK;
    if (!total_counts) return fail(PIXO_ERR_COMPRESSION, "Compression error: optimised tables need the statistics of all bands");
    uint64_t dc[2][12], ac[2][256];
    split_counts(total_counts, dc, ac);
    h = pixo_host::HuffSet::optimized(dc, ac, o.color_type != PIXO_GRAY);
    return PIXO_OK;
}
} // namespace

int pixo_hip_jpeg_splice(const pixo_jpeg_options *options, const uint64_t *total_counts, const uint8_t *const *pieces,
                         const size_t *piece_lens, uint32_t parts, uint8_t **out, size_t *out_len)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(pieces);
    PIXO_REQUIRE(piece_lens);
    PIXO_REQUIRE(out);
    PIXO_REQUIRE(out_len);
    std::string msg;
    int rc = pixo_host::validate(*options, false, 0, msg);
    if (rc) return fail(rc, msg);
    pixo_host::HuffSet h;
    if ((rc = tables_for_splice(*options, total_counts, h))) return rc;
    std::vector<uint8_t> v;
    if ((rc = pixo_host::splice_file(*options, h, pieces, piece_lens, parts, v, msg))) return fail(rc, msg);
    return hand_over(v, out, out_len);
}

int pixo_hip_jpeg_splice_layout(const pixo_jpeg_options *options, const uint64_t *total_counts, const uint8_t *piece_headers,
                                uint32_t parts, size_t *file_len, size_t *body_offsets)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(piece_headers);
    PIXO_REQUIRE(file_len);
    PIXO_REQUIRE(body_offsets);
    std::string msg;
    int rc = pixo_host::validate(*options, false, 0, msg);
    if (rc) return fail(rc, msg);
    pixo_host::HuffSet h;
    if ((rc = tables_for_splice(*options, total_counts, h))) return rc;
    pixo_host::SpliceLayout l;
    if ((rc = pixo_host::splice_layout(*options, h, piece_headers, parts, l, msg))) return fail(rc, msg);
    *file_len = l.file_len;
    for (uint32_t k = 0; k < parts; ++k) body_offsets[k] = l.body_off[k];
    return PIXO_OK;
}

int pixo_hip_jpeg_splice_finish(const pixo_jpeg_options *options, const uint64_t *total_counts, const uint8_t *piece_headers,
                                uint32_t parts, uint8_t *file, size_t file_len)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(piece_headers);
    PIXO_REQUIRE(file);
    std::string msg;
    int rc = pixo_host::validate(*options, false, 0, msg);
    if (rc) return fail(rc, msg);
    pixo_host::HuffSet h;
    if ((rc = tables_for_splice(*options, total_counts, h))) return rc;
    pixo_host::SpliceLayout l;
    if ((rc = pixo_host::splice_layout(*options, h, piece_headers, parts, l, msg))) return fail(rc, msg);
    if (file_len != l.file_len)
        return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(l.file_len) + " bytes");
    pixo_host::splice_finish(l, file);
    return PIXO_OK;
}

// ---- host twins of the band encoder (a band's tuple in host memory) -----------------------------------
namespace {
int band_options(const pixo_jpeg_options *options, uint32_t band_rows_, pixo_jpeg_options *band)
{
    std::string msg;
    int rc = pixo_host::validate(*options, false, 0, msg);
    if (rc) return fail(rc, msg);
    if (band_rows_ == 0 || band_rows_ > options->height) return fail(PIXO_ERR_COMPRESSION, "Compression error: bad band height");
    *band = *options;
    band->height = band_rows_;
    band->has_restart_interval = 0; band->restart_interval = 0;
    return PIXO_OK;
}
} // namespace

int pixo_hip_jpeg_band_count_host(const int16_t *y, const int16_t *cb, const int16_t *cr, const pixo_jpeg_options *options,
                                  uint32_t band_rows_, const int16_t prev_dc[3], uint64_t counts[PIXO_HIP_COUNT_WORDS])
{
    PIXO_REQUIRE(options); PIXO_REQUIRE(y); PIXO_REQUIRE(prev_dc); PIXO_REQUIRE(counts);
    pixo_jpeg_options band;
    int rc = band_options(options, band_rows_, &band);
    if (rc) return rc;
    uint64_t dc[2][12], ac[2][256];
    pixo_host::band_histograms(y, cb, cr, band, prev_dc, dc, ac);
    for (int cls = 0; cls < 2; ++cls) {
        std::memcpy(counts + cls * 268, dc[cls], sizeof dc[cls]);
        std::memcpy(counts + cls * 268 + 12, ac[cls], sizeof ac[cls]);
    }
    return PIXO_OK;
}

int pixo_hip_jpeg_band_bits_host(const int16_t *y, const int16_t *cb, const int16_t *cr, const pixo_jpeg_options *options,
                                 uint32_t band_rows_, const int16_t prev_dc[3], const uint64_t *total_counts, uint64_t *bits)
{
    PIXO_REQUIRE(options); PIXO_REQUIRE(y); PIXO_REQUIRE(prev_dc); PIXO_REQUIRE(bits);
    pixo_jpeg_options band;
    int rc = band_options(options, band_rows_, &band);
    if (rc) return rc;
    pixo_host::HuffSet h;
    if ((rc = tables_for_splice(*options, total_counts, h))) return rc;
    *bits = pixo_host::band_bits(y, cb, cr, band, h, prev_dc);
    return PIXO_OK;
}

int pixo_hip_jpeg_band_piece_host(const int16_t *y, const int16_t *cb, const int16_t *cr, const pixo_jpeg_options *options,
                                  uint32_t band_rows_, const int16_t prev_dc[3], const uint64_t *total_counts, uint64_t bit_offset,
                                  uint8_t **piece, size_t *piece_len)
{
    PIXO_REQUIRE(options); PIXO_REQUIRE(y); PIXO_REQUIRE(prev_dc); PIXO_REQUIRE(piece); PIXO_REQUIRE(piece_len);
    pixo_jpeg_options band;
    int rc = band_options(options, band_rows_, &band);
    if (rc) return rc;
    pixo_host::HuffSet h;
    if ((rc = tables_for_splice(*options, total_counts, h))) return rc;
    std::vector<uint8_t> v;
    pixo_host::band_piece(y, cb, cr, band, h, prev_dc, bit_offset, v);
    return hand_over(v, piece, piece_len);
}

// ---- the whole exchange inside one process: one thread per band/device ---------------------------------
namespace {
class PhaseBarrier { // every band thread arrives at every phase boundary, also after a failure
  public:
    explicit PhaseBarrier(unsigned n) : n_(n) {}
    void arrive()
    {
        std::unique_lock<std::mutex> lock(m_);
        const unsigned gen = gen_;
        if (++count_ == n_) { count_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(lock, [&] { return gen_ != gen; });
    }
  private:
    std::mutex m_;
    std::condition_variable cv_;
    unsigned n_, count_ = 0, gen_ = 0;
};
} // namespace

int pixo_hip_jpeg_encode_multi(const uint8_t *data, size_t data_len, const pixo_jpeg_options *options, const int *devices,
                               uint32_t n_devices, uint8_t **out, size_t *out_len)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(out);
    PIXO_REQUIRE(out_len);
    PIXO_REQUIRE(devices);
    const pixo_jpeg_options &o = *options;
    std::string msg;
    int rc = pixo_host::validate(o, true, data_len, msg);
    if (rc) return fail(rc, msg);
    PIXO_REQUIRE(data);
    if (n_devices == 0 || n_devices > 1024) return fail(PIXO_ERR_COMPRESSION, "Compression error: need 1..1024 devices");
    const pixo_host::Geometry whole = pixo_host::geometry(o.width, o.height, o.color_type, o.subsampling);
    if (!band_codable(o, whole)) { // progressive scans / restart markers: one device codes the whole tuple
        DeviceScope on(devices[0]);
        if (on.err != hipSuccess) return hip_fail(on.err, "hipSetDevice");
        const int keep = t_slot.device;
        if ((rc = pixo_hip_set_device(devices[0]))) return rc;
        rc = pixo_hip_jpeg_encode(data, data_len, options, out, out_len);
        (void)pixo_hip_set_device(keep);
        return rc;
    }
    const uint32_t parts = n_devices;
    const size_t bpp = whole.gray ? 1 : 3;
    struct Band {
        pixo_hip_band_encoder *enc = nullptr;
        int16_t last_dc[3] = {0, 0, 0}, prev_dc[3] = {0, 0, 0};
        uint64_t counts[PIXO_HIP_COUNT_WORDS];
        uint64_t bits = 0;
        int rc = PIXO_OK;
        std::string error;
    };
    std::vector<Band> bands(parts);
    std::vector<uint64_t> total_counts(PIXO_HIP_COUNT_WORDS, 0);
    std::vector<uint8_t> headers(static_cast<size_t>(parts) * pixo_host::kPieceHeader, 0);
    pixo_host::SpliceLayout layout;
    uint8_t *file = nullptr;
    PhaseBarrier barrier(parts);
    std::atomic<bool> failed{false};
    auto body = [&](unsigned k) {
        Band &b = bands[k];
        auto step = [&](int r) { if (r && !b.rc) { b.rc = r; b.error = t_error; failed.store(true); } };
        step(pixo_hip_band_encoder_create(options, parts, k, devices[k], &b.enc));
        if (b.enc) step(pixo_hip_band_encoder_coeffs(b.enc, data + static_cast<size_t>(b.enc->row_begin) * o.width * bpp, 0, b.last_dc));
        barrier.arrive(); // ---- exchange 1: the DCs at the band boundaries (3 x i16 per band)
        if (!failed.load()) {
            for (unsigned j = 0; j < k; ++j) // predictors = last DCs of the nearest band above that has rows
                if (bands[j].enc->rows) std::memcpy(b.prev_dc, bands[j].last_dc, sizeof b.prev_dc);
            if (o.optimize_huffman) step(pixo_hip_band_encoder_count(b.enc, b.prev_dc, b.counts));
        }
        if (o.optimize_huffman) {
            barrier.arrive(); // ---- exchange 1b: symbol statistics, summed (536 x u64 per band)
            if (k == 0 && !failed.load())
                for (unsigned j = 0; j < parts; ++j)
                    for (int i = 0; i < PIXO_HIP_COUNT_WORDS; ++i) total_counts[i] += bands[j].counts[i];
            barrier.arrive();
        }
        const uint64_t *tc = o.optimize_huffman ? total_counts.data() : nullptr;
        if (!failed.load()) step(pixo_hip_band_encoder_lengths(b.enc, b.prev_dc, tc, &b.bits));
        barrier.arrive(); // ---- exchange 2: bits per band (u64 per band) -> every band's bit offset
        if (!failed.load()) {
            uint64_t off = 0;
            for (unsigned j = 0; j < k; ++j) off += bands[j].bits;
            step(pixo_hip_band_encoder_pack_device(b.enc, off, headers.data() + static_cast<size_t>(k) * pixo_host::kPieceHeader, nullptr, nullptr));
        }
        barrier.arrive(); // ---- exchange 3: the 16-byte piece headers -> where every body goes in the file
        if (k == 0 && !failed.load()) {
            pixo_host::HuffSet h;
            std::string m;
            int r = tables_for_splice(o, tc, h);
            if (!r && (r = pixo_host::splice_layout(o, h, headers.data(), parts, layout, m))) r = fail(r, m);
            if (!r && !(file = static_cast<uint8_t *>(std::malloc(layout.file_len)))) r = fail(PIXO_ERR_COMPRESSION, "Compression error: out of host memory");
            step(r);
        }
        barrier.arrive();
        // every band's bytes go straight to their final place, over its own GPU's PCIe link, on its own thread
        if (!failed.load()) step(pixo_hip_band_encoder_copy_body(b.enc, file + layout.body_off[k]));
        pixo_hip_band_encoder_destroy(b.enc);
        b.enc = nullptr;
    };
    run_on_threads(parts, body);
    for (Band &b : bands)
        if (b.rc) { const int r = b.rc; const std::string e = b.error; std::free(file); return fail(r, e); }
    pixo_host::splice_finish(layout, file);
    *out = file;
    *out_len = layout.file_len;
    return PIXO_OK;
}

int pixo_hip_band(uint32_t width, uint32_t height, uint8_t color_type, uint8_t subsampling, uint32_t parts,
                  uint32_t index, uint32_t *row_begin, uint32_t *row_end, size_t *y_offset, size_t *y_blocks,
                  size_t *c_offset, size_t *c_blocks)
{
    if (width == 0 || height == 0)
        return fail(PIXO_ERR_INVALID_DIMENSIONS,
                    "Invalid image dimensions: " + std::to_string(width) + "x" + std::to_string(height));
    if (parts == 0 || index >= parts) return fail(PIXO_ERR_COMPRESSION, "Compression error: bad band index");
    const pixo_host::Geometry g = pixo_host::geometry(width, height, color_type, subsampling);
    const uint32_t unit_px = g.s420 ? 16 : 8;
    // contiguous unit-row bands, the first (units_y % parts) bands one row taller
    const uint32_t base = g.units_y / parts, extra = g.units_y % parts;
    const uint32_t u0 = index * base + (index < extra ? index : extra);
    const uint32_t u1 = u0 + base + (index < extra ? 1 : 0);
    *row_begin = u0 * unit_px < height ? u0 * unit_px : height;
    *row_end = u1 * unit_px < height ? u1 * unit_px : height;
    const size_t per_row_y = static_cast<size_t>(g.units_x) * (g.s420 ? 4 : 1);
    *y_offset = u0 * per_row_y;
    *y_blocks = (u1 - u0) * per_row_y;
    *c_offset = g.gray ? 0 : static_cast<size_t>(u0) * g.units_x;
    *c_blocks = g.gray ? 0 : static_cast<size_t>(u1 - u0) * g.units_x;
    return PIXO_OK;
}

int pixo_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pixo_hip_set_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n)
        return fail(PIXO_ERR_COMPRESSION, "Compression error: no HIP device " + std::to_string(device));
    if (t_slot.c && t_slot.c->device != device) { // rebind: park the old device's context, adopt one of the new device
        pool().give(t_slot.c);
        t_slot.c = nullptr;
    }
    t_slot.device = device;
    return PIXO_OK;
}

int pixo_hip_set_producer_stream(void *stream)
{
    t_producer = static_cast<hipStream_t>(stream);
    return PIXO_OK;
}

int pixo_hip_trim(void)
{
    if (t_slot.c) t_slot.c->release();
    pool().drain();
    return PIXO_OK;
}

void pixo_hip_free(void *p) { std::free(p); }

const char *pixo_hip_last_error(void) { return t_error.c_str(); }

const char *pixo_hip_version(void) { return "pixo_hip 0.1.0 (gfx950; reference pixo 0.4.1)"; }

} // extern "C"
