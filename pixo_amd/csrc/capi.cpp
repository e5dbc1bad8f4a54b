// capi.cpp — the extern "C" boundary declared in include/pixo_hip.h.
//
// Owns the thread-local device context (HIP stream, grow-only device and pinned host
// buffers) and the per-device quantiser-table cache.  No CPU fallback exists: without a
// usable GPU every compute entry point fails with PIXO_ERR_COMPRESSION and says so.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

// ---- thread-local execution context --------------------------------------------------
std::atomic<bool> g_exiting{false};
std::once_flag g_exit_hook;
struct Context {
    int device = 0;
    bool ready = false;
    hipStream_t stream = nullptr;
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
    Buf e_tables, e_hist, e_len, e_off, e_tmp, e_totals, e_stream, e_tile_ff, e_tile_base, e_out, e_seg_bytes, e_seg_off;
    Buf p_in, p_out, p_sums, p_scratch; // PNG filter stage
    Buf t_raw, t_trail;                 // progressive + trellis: unquantised DCT blocks (f32), Viterbi back-pointers
    Buf g_flags, g_rank, g_by_rank;     // progressive scans: band flags, rank among non-empty blocks and its inverse
    unsigned long long *h_sums = nullptr; size_t hsums_cap = 0; // pinned
    uint64_t *h_totals = nullptr; // pinned, 2 words
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
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        std::call_once(g_exit_hook, [] { std::atexit([] { g_exiting.store(true); }); });
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
    // A thread that ends gives its buffers back (servers with a thread per request would otherwise run the
    // device out of memory).  Not during process exit: an atexit handler registered at first use — it runs
    // before the HIP runtime's own, which were registered when the runtime was loaded — raises a flag, and
    // from then on everything is left to the dying process.
    ~Context();
    void release(); // everything back to the driver; the context starts over at its next use
};
void Context::release()
{
    if (!ready) return;
    if (hipSetDevice(device) != hipSuccess) return;
    if (stream) (void)hipStreamSynchronize(stream);
    Buf *bufs[] = {&e_tables, &e_hist, &e_len, &e_off, &e_tmp, &e_totals, &e_stream, &e_tile_ff, &e_tile_base, &e_out, &e_seg_bytes,
                   &e_seg_off, &p_in, &p_out, &p_sums, &p_scratch, &t_raw, &t_trail, &g_flags, &g_rank, &g_by_rank};
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
    d_px = d_coef = h_coef = nullptr; px_cap = coef_cap = hcoef_cap = 0;
    h_sums = nullptr; hsums_cap = 0; h_totals = nullptr; h_file = nullptr; hfile_cap = 0;
    stream = nullptr; ready = false;
}
Context::~Context()
{
    // (PIXO_HIP_KEEP_ON_THREAD_EXIT=1: diagnostics — leave a finished thread's buffers to the process)
    static const bool keep = std::getenv("PIXO_HIP_KEEP_ON_THREAD_EXIT") != nullptr;
    if (!keep && !g_exiting.load()) release();
}
thread_local Context t_ctx;

// Runs the device pipeline for host pixels; on success `*coef` points at pinned host
// memory holding [y | cb | cr] contiguously.
int coeffs_to_pinned(const uint8_t *pixels, const pixo_jpeg_options &o, const pixo_host::Geometry &g,
                     const int16_t **y, const int16_t **cb, const int16_t **cr)
{
    Context &c = t_ctx;
    int rc = c.ensure();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c.device));
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
int coeffs_on_device(const void *d_pixels, const pixo_jpeg_options &o, const pixo_host::Geometry &g, hipStream_t stream,
                     int16_t **dy, int16_t **dcb, int16_t **dcr)
{
    Context &c = t_ctx;
    const float *qt_all = nullptr;
    int rc = device_tables(c.device, &qt_all);
    if (rc) return rc;
    const size_t coef_bytes = (g.y_blocks + 2 * g.c_blocks) * 128;
    if ((rc = c.reserve_coef(coef_bytes))) return rc;
    *dy = static_cast<int16_t *>(c.d_coef);
    *dcb = *dy + g.y_blocks * 64;
    *dcr = *dcb + g.c_blocks * 64;
    HIP_TRY(pixo_dev::launch_jpeg_coeffs(d_pixels, o.width, o.height, g.gray, g.s420, 1, *dy,
                                         g.gray ? nullptr : *dcb, g.gray ? nullptr : *dcr,
                                         qt_all + (o.quality - 1) * pixo_host::kDeviceQtFloats, stream));
    return PIXO_OK;
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

// Device coefficient tuple -> whole file in the context's PINNED host buffer (headers written by
// the host, entropy-coded segment by the kernels of jpeg_entropy.hip and copied straight behind
// them).  Pinned on purpose: a device-to-host copy into fresh pageable memory makes the runtime
// pin those pages first, which costs 10-25 ms for an 11 MB file every time the address changes.
// batch > 1 (standard tables, no restart markers): the tuples of `batch` equal images back to back;
// every image is a byte-aligned segment of ONE packed stream.  Then *file = headers (once) followed by
// all the entropy-coded segments, and image_starts[i] (batch + 1 entries) are their offsets behind the
// headers; no EOI is written.
int device_entropy_to_pinned(const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
                             const pixo_host::Geometry &g, hipStream_t stream, const uint8_t **file, size_t *file_len,
                             uint32_t batch = 1, std::vector<uint64_t> *image_starts = nullptr, size_t *header_len = nullptr,
                             uint8_t *dest = nullptr, size_t dest_cap = 0)
{ // dest != null: the file goes straight into the caller's storage (no pinned intermediate); when it does not
  // fit, *file_len says how much is needed and nothing is copied (PIXO_ERR_BUFFER_TOO_SMALL)
    Stopwatch sw;
    Context &c = t_ctx;
    namespace pd = pixo_dev;
    const uint64_t n = (g.y_blocks + 2 * g.c_blocks) * batch;
    pd::ScanArgs a;
    a.y = dy; a.cb = dcb; a.cr = dcr;
    a.mode = g.gray ? 0 : (g.s420 ? 2 : 1);
    a.nblocks = n;
    a.blocks_per_mcu = g.gray ? 1 : (g.s420 ? 6 : 3);
    a.marker_bytes = 2;
    a.restart = scan_has_restart_markers(o, g) ? o.restart_interval : 0;
    uint64_t nseg = a.restart ? (g.units + a.restart - 1) / a.restart : 0;
    if (batch > 1) { // one segment per image, no marker between them
        a.restart = static_cast<uint32_t>(g.units);
        a.marker_bytes = 0;
        nseg = batch;
    }
    HIP_TRY(c.e_tables.reserve(pixo_host::kScanTableWords * 4));
    HIP_TRY(c.e_hist.reserve(pixo_host::kScanTableWords * 8));
    HIP_TRY(c.e_len.reserve(n * 4));
    HIP_TRY(c.e_off.reserve(n * 8));
    // scratch of the three prefix sums (blocks, restart segments, 0xFF tiles), reserved before any launch:
    // a block has at most 1665 bits, so the packed stream has at most n * 209 + 3 * nseg bytes
    const size_t tmp_blocks = pd::scan_tile_count(n) + 1, tmp_segs = pd::scan_tile_count(nseg ? nseg : 1) + 1;
    const size_t tmp_tiles = pd::scan_tile_count(pd::stuff_tile_count(n * 209 + 3 * nseg + 8)) + 1;
    HIP_TRY(c.e_tmp.reserve((tmp_blocks + tmp_segs + tmp_tiles) * 8));
    HIP_TRY(c.e_totals.reserve(16));
    if (!c.h_totals) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c.h_totals), 16, hipHostMallocDefault));
    a.tables = c.e_tables.as<uint32_t>();
    sw.lap("  reserve");

    pixo_host::HuffSet h;
    if (o.optimize_huffman) { // count_block statistics on the device, table construction on the host
        HIP_TRY(hipMemsetAsync(c.e_hist.p, 0, pixo_host::kScanTableWords * 8, stream));
        HIP_TRY(pd::launch_scan_count(a, c.e_hist.as<unsigned long long>(), stream));
        uint64_t counts[pixo_host::kScanTableWords];
        HIP_TRY(hipMemcpyAsync(counts, c.e_hist.p, sizeof counts, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        uint64_t dc[2][12], ac[2][256];
        for (int cls = 0; cls < 2; ++cls) {
            std::memcpy(dc[cls], counts + cls * 268, sizeof dc[cls]);
            std::memcpy(ac[cls], counts + cls * 268 + 12, sizeof ac[cls]);
        }
        h = pixo_host::HuffSet::optimized(dc, ac, !g.gray);
    } else {
        h = pixo_host::HuffSet::standard();
    }
    uint32_t packed[pixo_host::kScanTableWords];
    pixo_host::pack_scan_tables(h, packed);
    HIP_TRY(hipMemcpyAsync(c.e_tables.p, packed, sizeof packed, hipMemcpyHostToDevice, stream));
    sw.lap("  tables h2d");

    // 1-2: block bit lengths, their prefix sum
    HIP_TRY(pd::launch_scan_lengths(a, c.e_len.as<uint32_t>(), stream));
    HIP_TRY(pd::launch_exclusive_scan(c.e_len.as<uint32_t>(), n, c.e_off.as<uint64_t>(), c.e_tmp.as<uint64_t>(),
                                      c.e_totals.as<uint64_t>(), stream));
    pd::SegmentPlan plan{0, nullptr};
    if (nseg) { // restart markers: byte-aligned segments, each followed by two marker bytes
        HIP_TRY(c.e_seg_bytes.reserve(nseg * 8));
        HIP_TRY(c.e_seg_off.reserve(nseg * 8));
        HIP_TRY(pd::launch_segment_sizes(a, c.e_off.as<uint64_t>(), c.e_totals.as<uint64_t>(), nseg, c.e_seg_bytes.as<uint32_t>(), stream));
        HIP_TRY(pd::launch_exclusive_scan(c.e_seg_bytes.as<uint32_t>(), nseg, c.e_seg_off.as<uint64_t>(),
                                          c.e_tmp.as<uint64_t>() + tmp_blocks, c.e_totals.as<uint64_t>() + 1, stream));
        plan.nsegments = nseg;
        plan.seg_byte_off = c.e_seg_off.as<uint64_t>();
    }
    HIP_TRY(hipMemcpyAsync(c.h_totals, c.e_totals.p, 16, hipMemcpyDeviceToHost, stream));
    sw.lap("  launches");
    HIP_TRY(hipStreamSynchronize(stream)); // `packed` may go out of scope after this, too
    sw.lap("tables+lengths+scan");
    const uint64_t total_bits = c.h_totals[0];
    const uint64_t nbytes = nseg ? c.h_totals[1] : (total_bits + 7) / 8; // bytes of the packed (unstuffed) stream
    // 3: pack
    const size_t stream_bytes = (nbytes / 4 + 2) * 4;
    HIP_TRY(c.e_stream.reserve(stream_bytes));
    HIP_TRY(hipMemsetAsync(c.e_stream.p, 0, stream_bytes, stream));
    HIP_TRY(pd::launch_scan_pack(a, c.e_off.as<uint64_t>(), total_bits, nseg ? &plan : nullptr, c.e_stream.as<uint32_t>(), stream));
    // 4: 0xFF census
    const size_t tiles = pd::stuff_tile_count(nbytes);
    HIP_TRY(c.e_tile_ff.reserve(tiles * 4));
    HIP_TRY(c.e_tile_base.reserve(tiles * 8));
    HIP_TRY(pd::launch_ff_tile_count(c.e_stream.as<uint32_t>(), nbytes, c.e_tile_ff.as<uint32_t>(), stream));
    HIP_TRY(pd::launch_exclusive_scan(c.e_tile_ff.as<uint32_t>(), tiles, c.e_tile_base.as<uint64_t>(),
                                      c.e_tmp.as<uint64_t>() + tmp_blocks + tmp_segs, c.e_totals.as<uint64_t>() + 1, stream));
    HIP_TRY(hipMemcpyAsync(c.h_totals + 1, c.e_totals.as<uint64_t>() + 1, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    sw.lap("memset+pack+ff census");
    const uint64_t scan_bytes = nbytes + c.h_totals[1];
    // 5: stuff, then straight into the caller's vector behind the headers
    HIP_TRY(c.e_out.reserve(scan_bytes));
    HIP_TRY(pd::launch_stuff(c.e_stream.as<uint32_t>(), nbytes, c.e_tile_base.as<uint64_t>(), c.e_out.as<uint8_t>(), stream));
    if (nseg) HIP_TRY(pd::launch_restart_markers(a, c.e_off.as<uint64_t>(), plan, c.e_stream.as<uint32_t>(), c.e_tile_base.as<uint64_t>(),
                                                 c.e_out.as<uint8_t>(), stream));
    if (batch > 1) { // where every image's segment begins in the stuffed stream (reuses the seg_bytes buffer: 8 B/entry)
        HIP_TRY(c.e_seg_bytes.reserve(nseg * 8));
        HIP_TRY(pd::launch_segment_out_offsets(plan, nbytes, c.e_stream.as<uint32_t>(), c.e_tile_base.as<uint64_t>(),
                                               c.e_seg_bytes.as<uint64_t>(), stream));
        image_starts->assign(batch + 1, 0);
        HIP_TRY(hipMemcpyAsync(image_starts->data(), c.e_seg_bytes.p, nseg * 8, hipMemcpyDeviceToHost, stream));
        (*image_starts)[batch] = scan_bytes;
    }
    std::vector<uint8_t> head;
    pixo_host::file_headers(head, o, h);
    const size_t hdr = head.size(), total = hdr + scan_bytes + 2;
    uint8_t *buf = dest;
    if (dest) {
        if (total > dest_cap) {
            *file_len = total;
            return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(total) + " bytes");
        }
    } else {
        int rc = c.reserve_hfile(total);
        if (rc) return rc;
        buf = c.h_file;
    }
    std::memcpy(buf, head.data(), hdr);
    HIP_TRY(hipMemcpyAsync(buf + hdr, c.e_out.p, scan_bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
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

int device_entropy_to_malloc(const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
                             const pixo_host::Geometry &g, hipStream_t stream, uint8_t **out_buf, size_t *out_len)
{
    const uint8_t *file = nullptr;
    size_t n = 0;
    int rc = device_entropy_to_pinned(dy, dcb, dcr, o, g, stream, &file, &n);
    if (rc) return rc;
    return deliver(file, n, out_buf, out_len);
}

// The seven scans of simple_progressive_script (progressive.rs:98-110) coded by the kernels of
// jpeg_entropy.hip over the device tuple; `out` already holds the file headers.  All scans are ONE
// packed stream of byte-aligned segments (the virtual block order is scan by scan, storage order inside
// a scan), so lengths / prefix sum / pack / 0xFF stuffing run once; the host only splices the seven SOS
// headers between the stuffed segments.
int device_progressive_scans(const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_host::Geometry &g,
                             const pixo_host::HuffSet &h, Context &c, std::vector<uint8_t> &out)
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
    HIP_TRY(c.e_tables.reserve(pixo_host::kScanTableWords * 4));
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
    if (!c.h_totals) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c.h_totals), 16, hipHostMallocDefault));
    a.tables = c.e_tables.as<uint32_t>();
    a.flags = c.g_flags.as<uint32_t>();
    a.nonempty = c.e_len.as<uint32_t>(); // only the input of the rank prefix sum: the lengths reuse it
    a.rank = c.g_rank.as<uint64_t>();
    a.by_rank = c.g_by_rank.as<uint32_t>();

    uint32_t packed[pixo_host::kScanTableWords];
    pixo_host::pack_scan_tables(h, packed);
    for (uint32_t &w : packed) // progressive.rs:363-381: a symbol the table lacks is coded as (0, 4 bits)
        if ((w >> 16) == 0) w = 4u << 16;
    HIP_TRY(hipMemcpyAsync(c.e_tables.p, packed, sizeof packed, hipMemcpyHostToDevice, stream));
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
    int rc = c.reserve_hfile(scan_bytes + 16);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c.h_file, c.e_out.p, scan_bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    start[7] = scan_bytes;
    for (int i = 6; i >= 0; --i)
        if (start[i] == ~0ull) start[i] = start[i + 1]; // empty scans at the end of the stream
    static const uint8_t script[7][3] = {{0, 0, 0}, {1, 0, 0}, {2, 0, 0}, {0, 1, 10}, {0, 11, 63}, {1, 1, 63}, {2, 1, 63}};
    out.reserve(out.size() + scan_bytes + 7 * 10 + 2);
    for (int i = 0; i < 7; ++i) { // write_sos_progressive, jpeg/mod.rs:650-682
        const uint8_t sos[10] = {0xFF, 0xDA, 0, 8, 1, static_cast<uint8_t>(script[i][0] + 1),
                                 static_cast<uint8_t>(script[i][0] == 0 ? 0x00 : 0x11), script[i][1], script[i][2], 0};
        out.insert(out.end(), sos, sos + 10);
        out.insert(out.end(), c.h_file + start[i], c.h_file + start[i + 1]);
    }
    out.push_back(0xFF); out.push_back(0xD9);
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
    HIP_TRY(c.e_hist.reserve(pixo_host::kScanTableWords * 8));
    HIP_TRY(hipMemsetAsync(c.e_hist.p, 0, pixo_host::kScanTableWords * 8, c.stream));
    HIP_TRY(pd::launch_scan_count(a, c.e_hist.as<unsigned long long>(), c.stream));
    uint64_t counts[pixo_host::kScanTableWords];
    HIP_TRY(hipMemcpyAsync(counts, c.e_hist.p, sizeof counts, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    uint64_t dc[2][12], ac[2][256];
    for (int cls = 0; cls < 2; ++cls) {
        std::memcpy(dc[cls], counts + cls * 268, sizeof dc[cls]);
        std::memcpy(ac[cls], counts + cls * 268 + 12, sizeof ac[cls]);
    }
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
int progressive_to_vector(const void *d_pixels, const pixo_jpeg_options &o, const pixo_host::Geometry &g, Context &c,
                          std::vector<uint8_t> &out)
{
    namespace pd = pixo_dev;
    int rc;
    int16_t *dy = nullptr, *dcb = nullptr, *dcr = nullptr;
    const float *qt_all = nullptr;
    if ((rc = device_tables(c.device, &qt_all))) return rc;
    const float *qt = qt_all + (o.quality - 1) * pixo_host::kDeviceQtFloats;
    pixo_host::HuffSet h;
    const bool need_plain = o.optimize_huffman || !o.trellis_quant;
    if (need_plain && (rc = coeffs_on_device(d_pixels, o, g, c.stream, &dy, &dcb, &dcr))) return rc;
    if ((rc = huffman_for_tuple(dy, dcb, dcr, o, g, c, h))) return rc;
    const size_t blocks = g.y_blocks + 2 * g.c_blocks, coef_bytes = blocks * 128;
    if (o.trellis_quant) {
        HIP_TRY(c.t_raw.reserve(blocks * 256));
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
        out.clear();
        pixo_host::file_headers(out, o, h);
        return device_progressive_scans(dy, dcb, dcr, g, h, c, out);
    }
    if ((rc = c.reserve_hcoef(coef_bytes))) return rc;
    HIP_TRY(hipMemcpyAsync(c.h_coef, dy, coef_bytes, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    const int16_t *hy = static_cast<const int16_t *>(c.h_coef), *hcb = hy + g.y_blocks * 64, *hcr = hcb + g.c_blocks * 64;
    pixo_host::encode_progressive_file(hy, hcb, hcr, o, h, out);
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
int encode_to_view(const uint8_t *data, size_t data_len, const pixo_jpeg_options &o, std::vector<uint8_t> &spill,
                   const uint8_t **file, size_t *file_len)
{
    std::string msg;
    int rc = pixo_host::validate(o, true, data_len, msg);
    if (rc) return fail(rc, msg);
    const pixo_host::Geometry g = pixo_host::geometry(o.width, o.height, o.color_type, o.subsampling);
    if (!o.progressive && std::getenv("PIXO_HIP_HOST_ENTROPY")) { // (experiments: the host twin of the entropy stage)
        const int16_t *y, *cb, *cr;
        if ((rc = coeffs_to_pinned(data, o, g, &y, &cb, &cr))) return rc;
        pixo_host::encode_file(y, cb, cr, o, spill);
        *file = spill.data();
        *file_len = spill.size();
        return PIXO_OK;
    }
    Context &c = t_ctx;
    if ((rc = c.ensure())) return rc;
    HIP_TRY(hipSetDevice(c.device));
    const size_t px_bytes = static_cast<size_t>(o.width) * o.height * (g.gray ? 1 : 3);
    if ((rc = c.reserve_px((px_bytes + 15) & ~size_t{15}))) return rc;
    HIP_TRY(hipMemcpyAsync(c.d_px, data, px_bytes, hipMemcpyHostToDevice, c.stream));
    if (o.progressive) {
        if ((rc = progressive_to_vector(c.d_px, o, g, c, spill))) return rc;
        *file = spill.data();
        *file_len = spill.size();
        return PIXO_OK;
    }
    int16_t *dy, *dcb, *dcr;
    if ((rc = coeffs_on_device(c.d_px, o, g, c.stream, &dy, &dcb, &dcr))) return rc;
    return device_entropy_to_pinned(dy, dcb, dcr, o, g, c.stream, file, file_len);
}

} // namespace

// A null pointer where the contract wants an object is a caller bug the Rust API cannot express; the C ABI
// answers it with an error instead of a crash.
#define PIXO_REQUIRE(p) do { if (!(p)) return fail(PIXO_ERR_COMPRESSION, "Compression error: null argument '" #p "'"); } while (0)

extern "C" {

void pixo_jpeg_options_from_preset(pixo_jpeg_options *o, uint32_t width, uint32_t height,
                                   uint8_t quality, uint8_t preset)
{ // jpeg/mod.rs:162-216
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
    int rc = encode_to_view(data, data_len, *options, spill, &file, &n);
    if (rc) return rc;
    return deliver(file, n, out, out_len);
}

int pixo_hip_jpeg_encode_into(uint8_t *output, size_t capacity, const uint8_t *data, size_t data_len,
                              const pixo_jpeg_options *options, size_t *out_len)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(out_len);
    std::vector<uint8_t> spill;
    const uint8_t *file = nullptr;
    size_t n = 0;
    int rc = encode_to_view(data, data_len, *options, spill, &file, &n);
    if (rc) return rc;
    *out_len = n;
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
    const int16_t *hy, *hcb, *hcr;
    if ((rc = coeffs_to_pinned(pixels, o, g, &hy, &hcb, &hcr))) return rc;
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

int pixo_hip_jpeg_entropy_encode(const int16_t *y, const int16_t *cb, const int16_t *cr,
                                 const pixo_jpeg_options *options, uint8_t **out, size_t *out_len)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(out);
    PIXO_REQUIRE(out_len);
    std::string msg;
    int rc = pixo_host::validate(*options, false, 0, msg);
    if (rc) return fail(rc, msg);
    std::vector<uint8_t> v;
    pixo_host::encode_file(y, cb, cr, *options, v);
    return hand_over(v, out, out_len);
}

namespace {
// binds the thread-local context to the HIP device that is current for the caller
int context_on_current_device(Context **out)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (t_ctx.ready && t_ctx.device != dev) (void)pixo_hip_set_device(dev);
    t_ctx.device = dev;
    int rc = t_ctx.ensure();
    if (rc) return rc;
    *out = &t_ctx;
    return PIXO_OK;
}

int device_tuple_to_malloc(const int16_t *dy, const int16_t *dcb, const int16_t *dcr, const pixo_jpeg_options &o,
                           const pixo_host::Geometry &g, Context &c, uint8_t **out, size_t *out_len)
{
    if (!std::getenv("PIXO_HIP_HOST_ENTROPY")) {
        if (!o.progressive) return device_entropy_to_malloc(dy, dcb, dcr, o, g, c.stream, out, out_len);
        pixo_host::HuffSet h;
        int rc = huffman_for_tuple(dy, dcb, dcr, o, g, c, h);
        if (rc) return rc;
        std::vector<uint8_t> v;
        pixo_host::file_headers(v, o, h);
        if ((rc = device_progressive_scans(dy, dcb, dcr, g, h, c, v))) return rc;
        return hand_over(v, out, out_len);
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
    Context *c = nullptr;
    if ((rc = context_on_current_device(&c))) return rc;
    const pixo_host::Geometry g = pixo_host::geometry(options->width, options->height, options->color_type, options->subsampling);
    return device_tuple_to_malloc(static_cast<const int16_t *>(d_y), static_cast<const int16_t *>(d_cb),
                                  static_cast<const int16_t *>(d_cr), *options, g, *c, out, out_len);
}

int pixo_hip_jpeg_encode_device(const void *d_pixels, const pixo_jpeg_options *options, uint8_t **out, size_t *out_len)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(out);
    PIXO_REQUIRE(out_len);
    std::string msg;
    int rc = pixo_host::validate(*options, false, 0, msg);
    if (rc) return fail(rc, msg);
    Context *c = nullptr;
    if ((rc = context_on_current_device(&c))) return rc;
    const pixo_host::Geometry g = pixo_host::geometry(options->width, options->height, options->color_type, options->subsampling);
    if (options->progressive) {
        std::vector<uint8_t> v;
        if ((rc = progressive_to_vector(d_pixels, *options, g, *c, v))) return rc;
        return hand_over(v, out, out_len);
    }
    int16_t *dy, *dcb, *dcr;
    if ((rc = coeffs_on_device(d_pixels, *options, g, c->stream, &dy, &dcb, &dcr))) return rc;
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
    if (options->progressive || std::getenv("PIXO_HIP_HOST_ENTROPY")) { // assembled on the host: copy if it fits
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
    int16_t *dy, *dcb, *dcr;
    if ((rc = coeffs_on_device(d_pixels, *options, g, c->stream, &dy, &dcb, &dcr))) return rc;
    const uint8_t *file = nullptr;
    // (a null output with capacity 0 is a size query)
    static uint8_t nowhere;
    return device_entropy_to_pinned(dy, dcb, dcr, *options, g, c->stream, &file, out_len, 1, nullptr, nullptr,
                                    output ? output : &nowhere, output ? capacity : 0);
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
    Context &c = t_ctx;
    if ((rc = c.ensure())) return rc;
    HIP_TRY(hipSetDevice(c.device));
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
            if ((rc = coeffs_on_device(static_cast<const uint8_t *>(d_pixels) + i * px_bytes, o, g, c->stream, &dy, &dcb, &dcr))) return release(rc);
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
    if ((rc = device_entropy_to_pinned(dy, dcb, dcr, o, g, c->stream, &blob, &blob_len, batch, &starts, &hdr))) return rc;
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
    Context &c = t_ctx;
    if (c.ready && c.device != device) c.release(); // rebind: drop the old device's buffers and start over
    c.device = device;
    return PIXO_OK;
}

int pixo_hip_trim(void)
{
    t_ctx.release();
    return PIXO_OK;
}

void pixo_hip_free(void *p) { std::free(p); }

const char *pixo_hip_last_error(void) { return t_error.c_str(); }

const char *pixo_hip_version(void) { return "pixo_hip 0.1.0 (gfx950; reference pixo 0.4.1)"; }

} // extern "C"
