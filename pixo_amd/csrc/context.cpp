// context.cpp — error state, per-device quantiser-table cache, the thread's execution context (HIP stream, grow-only
// device and pinned host buffers), the pool in which contexts outlive their threads, device selection, debug switches.
#include <algorithm>

#include "capi_internal.hpp"
#include "jpeg_kernels.hpp"
#include "jpeg_trellis.hpp"

namespace pixo_capi {

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

namespace {
std::mutex g_qt_mutex;
float *g_qt[kMaxDevices] = {};
thread_local hipStream_t t_producer = nullptr;
} // namespace

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

// ---- debug switches ------------------------------------------------------------------------------------------------
namespace {
DebugSwitches parse_switches(const char *e)
{
    DebugSwitches v;
    if (!e) return v;
    const std::string all = e;
    for (size_t i = 0; i < all.size();) {
        size_t k = all.find(',', i);
        if (k == std::string::npos) k = all.size();
        const std::string item = all.substr(i, k - i);
        i = k + 1;
        const size_t eq = item.find('=');
        const std::string name = item.substr(0, eq), val = eq == std::string::npos ? "" : item.substr(eq + 1);
        const long num = val.empty() ? 0 : std::atol(val.c_str());
        if (name == "trace") v.trace = true;
        else if (name == "host_entropy") v.host_entropy = true;
        else if (name == "multipass_entropy") v.multipass_entropy = true;
        else if (name == "direct_stores") v.direct_stores = true;
        else if (name == "one_piece") v.one_piece = true;
        else if (name == "no_bands_upload") v.no_bands_upload = true;
        else if (name == "plain_host") v.plain_host = true;
        else if (name == "no_direct_small") v.no_direct_small = true;
        else if (name == "two_kernel_scan") v.two_kernel_scan = true;
        else if (name == "fused_batch") v.fused_batch = true;
        else if (name == "no_side_stats") v.no_side_stats = true;
        else if (name == "trellis_form") v.trellis_form = val == "lane" ? 1 : (val == "group" ? 2 : 0);
        else if (name == "coef_form") v.coef_form = val == "scalar" ? 1 : (val == "packed" ? 2 : 0);
        else if (name == "bands_upload_min_mb" && num > 0) v.bands_upload_min_mb = static_cast<uint32_t>(num);
        else if (name == "bands_upload_mb" && num > 0) v.bands_upload_mb = static_cast<uint32_t>(num);
        else if (name == "piece_groups" && num > 0) v.piece_groups = static_cast<uint64_t>(num);
        else if (name == "piece_medium") { v.piece_medium_forced = true; if (num > 0) v.piece_medium = static_cast<uint64_t>(num); }
        else if (name == "copy_threads") v.copy_threads = static_cast<unsigned>(num < 1 ? 1 : (num > 64 ? 64 : num));
        else if (name == "batch_parts" && num > 0) v.batch_parts = static_cast<uint32_t>(num);
        else if (name == "spin_budget" && !val.empty()) v.spin_budget = static_cast<uint32_t>(num < 0 ? 0 : num);
        else if (name == "piece_schedule") {
            std::vector<uint32_t> w;
            for (size_t p = 0; p < val.size();) {
                size_t q = val.find(':', p);
                if (q == std::string::npos) q = val.size();
                const long x = std::atol(val.substr(p, q - p).c_str());
                if (x > 0) w.push_back(static_cast<uint32_t>(x));
                p = q + 1;
            }
            if (!w.empty()) v.piece_schedule = w;
        } else if (!name.empty()) {
            std::fprintf(stderr, "[pixo_hip] PIXO_HIP_DEBUG: unknown switch '%s' ignored\n", name.c_str());
        }
    }
    return v;
}
DebugSwitches &switches()
{
    static DebugSwitches *d = [] {
        DebugSwitches *p = new DebugSwitches(parse_switches(std::getenv("PIXO_HIP_DEBUG")));
        pixo_dev::set_coef_form(p->coef_form);
        pixo_dev::set_trellis_form(p->trellis_form);
        return p;
    }();
    return *d;
}
} // namespace
const DebugSwitches &debug() { return switches(); }

// ---- Context ---------------------------------------------------------------------------------------------------------
int Context::ensure()
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
int Context::reserve_px(size_t n)
{
    if (n <= px_cap) return PIXO_OK;
    if (d_px) (void)hipFree(d_px);
    d_px = nullptr; px_cap = 0;
    HIP_TRY(hipMalloc(&d_px, n));
    px_cap = n;
    return PIXO_OK;
}
int Context::reserve_coef(size_t n)
{
    if (n > coef_cap) {
        if (d_coef) (void)hipFree(d_coef);
        d_coef = nullptr; coef_cap = 0;
        HIP_TRY(hipMalloc(&d_coef, n));
        coef_cap = n;
    }
    return PIXO_OK;
}
int Context::reserve_hcoef(size_t n)
{
    if (n > hcoef_cap) {
        if (h_coef) (void)hipHostFree(h_coef);
        h_coef = nullptr; hcoef_cap = 0;
        HIP_TRY(hipHostMalloc(&h_coef, n, hipHostMallocDefault));
        hcoef_cap = n;
    }
    return PIXO_OK;
}
int Context::reserve_hfile(size_t n)
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
int Context::reserve_hsegs(size_t words)
{
    if (words > hsegs_cap) {
        if (h_segs) (void)hipHostFree(h_segs);
        h_segs = nullptr; hsegs_cap = 0;
        // (never below 1024 words: the block also receives the 536 symbol counters of an optimised-tables pass, and a job that has handed
        // the block's address to a kernel — SegArgs::host_out_end — must not see it move when the counters ask for their room: a
        // standard-tables call followed by an optimised-tables call with restart intervals had the stuffing kernel write the segments'
        // ends into the freed block — a GPU memory fault, found by tools/stress_parity.py seed 955)
        const size_t want = (words < 1024 ? 1024 : words) + words / 4 + 16;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_segs), want * 8, hipHostMallocDefault));
        hsegs_cap = want;
    }
    return PIXO_OK;
}
int Context::ensure_totals()
{
    if (!h_totals) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_totals), kTotalsWords * 8, hipHostMallocDefault));
    return PIXO_OK;
}

namespace {
template <class F> void each_buf(Context &c, F &&f)
{
    Context::Buf *bufs[] = {&c.e_tables, &c.e_hist, &c.e_count, &c.e_len, &c.e_off, &c.e_tmp, &c.e_totals, &c.e_stream, &c.e_tile_ff, &c.e_tile_base,
                            &c.e_out, &c.e_seg_bytes, &c.e_seg_off, &c.e_code_state, &c.e_stuff_state, &c.e_pc_state, &c.e_pc_spill, &c.e_chain, &c.e_segs, &c.e_seams, &c.p_in, &c.p_out,
                            &c.p_sums, &c.p_scratch, &c.t_raw, &c.t_trail, &c.t_plain, &c.g_flags, &c.g_rank, &c.g_by_rank};
    for (Context::Buf *b : bufs) f(*b);
}
} // namespace

size_t Context::held_bytes() const
{
    size_t n = px_cap + coef_cap + hcoef_cap + hfile_cap + hsums_cap + hsegs_cap * 8;
    each_buf(const_cast<Context &>(*this), [&](Buf &b) { n += b.cap; });
    return n;
}

void Context::shrink_to(size_t max_buffer_bytes)
{ // (a live thread on the context's device; the stream is idle)
    if (!ready) return;
    DeviceScope on(device);
    if (on.err != hipSuccess) return;
    if (stream) (void)hipStreamSynchronize(stream);
    each_buf(*this, [&](Buf &b) {
        if (b.cap > max_buffer_bytes) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    });
    if (e_tables.p == nullptr) tables_valid = false;
    if (e_code_state.p == nullptr) code_state_zero_words = 0;
    if (e_pc_state.p == nullptr) pc_half_words = 0;
    if (px_cap > max_buffer_bytes) { (void)hipFree(d_px); d_px = nullptr; px_cap = 0; }
    if (coef_cap > max_buffer_bytes) { (void)hipFree(d_coef); d_coef = nullptr; coef_cap = 0; }
    if (hcoef_cap > max_buffer_bytes) { (void)hipHostFree(h_coef); h_coef = nullptr; hcoef_cap = 0; }
    if (hfile_cap > max_buffer_bytes) { (void)hipHostFree(h_file); h_file = nullptr; hfile_cap = 0; }
}

void Context::release()
{
    if (helper) { destroy_copy_helper(helper); helper = nullptr; } // (stops and joins the context's second host thread)
    if (!ready) return;
    DeviceScope on(device);
    if (on.err != hipSuccess) return;
    if (stream) (void)hipStreamSynchronize(stream);
    each_buf(*this, [](Buf &b) {
        if (b.p) (void)hipFree(b.p);
        b.p = nullptr; b.cap = 0;
    });
    if (d_px) (void)hipFree(d_px);
    if (d_coef) (void)hipFree(d_coef);
    if (h_coef) (void)hipHostFree(h_coef);
    if (h_sums) (void)hipHostFree(h_sums);
    if (h_totals) (void)hipHostFree(h_totals);
    if (h_segs) (void)hipHostFree(h_segs);
    if (h_tables) (void)hipHostFree(h_tables);
    h_tables = nullptr; tables_valid = false;
    if (h_file) (void)hipHostFree(h_file);
    if (stream) (void)hipStreamDestroy(stream);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (upload_stream) (void)hipStreamDestroy(upload_stream);
    copy_stream = upload_stream = nullptr;
    for (hipEvent_t e : piece_done) (void)hipEventDestroy(e);
    piece_done.clear();
    for (hipEvent_t e : band_up) (void)hipEventDestroy(e);
    band_up.clear();
    if (producer_done) (void)hipEventDestroy(producer_done);
    producer_done = nullptr;
    if (stats_done) (void)hipEventDestroy(stats_done);
    if (side_ready) (void)hipEventDestroy(side_ready);
    stats_done = side_ready = nullptr;
    code_state_zero_words = 0;
    tables_valid = false;
    d_px = d_coef = h_coef = nullptr; px_cap = coef_cap = hcoef_cap = 0;
    h_sums = nullptr; hsums_cap = 0; h_totals = nullptr; h_segs = nullptr; hsegs_cap = 0; h_file = nullptr; hfile_cap = 0;
    stream = nullptr; ready = false;
}

// Contexts outlive the threads that use them.  A thread's context must not be torn down by a
// thread-local destructor: those run when the HIP runtime's own per-thread state may already be gone
// (it was created later, so it is destroyed earlier), and — for threads still winding down while main()
// returns — concurrently with the runtime's atexit teardown; hipFree / hipHostFree from there crashed
// (SIGSEGV in amd::Context::svmFree with 16 threads ending at once, profiles/r02_thread_exit_crash.txt).
// So a thread that ends only parks its context here (a mutex and a vector push, no HIP call); the next
// thread that needs one adopts it — buffers, stream and all, which also spares a server with a thread per
// request every hipMalloc.  What is parked is bounded by BYTES as well as by count: after one 16384x16384 image a
// context holds ~2 GB of HBM and ~0.4 GB of pinned host memory, and a server with a thread per request would park
// sixteen of those.  Live threads enforce the bound when they take a context (the oldest parked contexts first give up
// their large buffers, then go entirely); pixo_hip_trim() releases everything.  A context of another device is never
// re-bound on the adopting thread's hot path: a thread that finds none of its own device makes a new one.  Nothing is
// destroyed at process exit (the pool is leaked on purpose).
constexpr size_t kIdleContextsKept = 16;
constexpr size_t kIdleBytesKept = size_t{8} << 30;    // device + pinned bytes all parked contexts together may keep (3 % of the 288 GB; round 6: with
                                                      // 1 GiB four parked contexts of 4096x4096 files shrank — hipFree, a device-wide synchronisation each —
                                                      // under the feet of the threads that took them over: 217 -> 650-917 us per file, tools/mt_device_files.py)
constexpr size_t kIdleBufferKept = size_t{64} << 20;  // a parked context that must shrink keeps buffers up to this size
Context *ContextPool::take(int device)
{
    std::vector<Context *> excess, shrink;
    Context *c = nullptr;
    {
        std::lock_guard<std::mutex> lock(m);
        for (size_t i = idle.size(); i-- > 0;)
            if (idle[i]->device == device) { c = idle[i]; idle.erase(idle.begin() + static_cast<long>(i)); break; }
        while (idle.size() > kIdleContextsKept) { excess.push_back(idle.front()); idle.erase(idle.begin()); }
        size_t bytes = 0;
        for (Context *x : idle) bytes += x->held_bytes();
        for (size_t i = 0; i < idle.size() && bytes > kIdleBytesKept;) { // oldest first
            Context *x = idle[i];
            bytes -= x->held_bytes();
            idle.erase(idle.begin() + static_cast<long>(i));
            shrink.push_back(x);
        }
    }
    for (Context *x : excess) { x->release(); delete x; } // (a live thread: HIP calls are fine here)
    for (Context *x : shrink) {
        x->shrink_to(kIdleBufferKept);
        give(x); // (back at the young end: small now)
    }
    if (!c) { c = new Context; c->device = device; }
    return c;
}
void ContextPool::give(Context *c) // no HIP calls: may run in a thread-local destructor
{
    std::lock_guard<std::mutex> lock(m);
    idle.push_back(c);
}
void ContextPool::drain() // frees every parked context (pixo_hip_trim)
{
    std::vector<Context *> all;
    {
        std::lock_guard<std::mutex> lock(m);
        all.swap(idle);
    }
    for (Context *x : all) { x->release(); delete x; }
}
ContextPool &pool()
{
    static ContextPool *p = new ContextPool;
    return *p;
}
ThreadSlot::~ThreadSlot() { if (c) pool().give(c); }
thread_local ThreadSlot t_slot;
Context &thread_context()
{
    if (!t_slot.c) t_slot.c = pool().take(t_slot.device);
    return *t_slot.c;
}

// Device-pointer entry points run on the context's own stream.  What the caller enqueued before the call
// — on the stream it named with pixo_hip_set_producer_stream, by default the NULL stream — is ordered
// in front of it with an event (no host synchronisation).
int order_after_producer(Context &c)
{
    if (!c.producer_done) HIP_TRY(hipEventCreateWithFlags(&c.producer_done, hipEventDisableTiming));
    if (hipEventRecord(c.producer_done, t_producer) != hipSuccess) {
        // the named stream is not one of this context's device (a stale setting from work on another GPU): an event of
        // this device cannot be recorded on it — wait for that stream on the host instead of failing the call
        (void)hipGetLastError();
        HIP_TRY(hipStreamSynchronize(t_producer));
        return PIXO_OK;
    }
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

} // namespace pixo_capi

using namespace pixo_capi;

extern "C" {

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

void *pixo_hip_get_producer_stream(void) { return t_producer; }

int pixo_hip_debug_configure(const char *switches_or_null)
{ // tests and tools only; not synchronised with calls in flight on other threads
    switches() = parse_switches(switches_or_null ? switches_or_null : std::getenv("PIXO_HIP_DEBUG"));
    pixo_dev::set_coef_form(switches().coef_form);
    pixo_dev::set_trellis_form(switches().trellis_form);
    return PIXO_OK;
}

int pixo_hip_trim(void)
{
    if (t_slot.c) t_slot.c->release();
    pool().drain();
    drop_kept_blocks();
    drop_batch_worker_buffers();
    return PIXO_OK;
}

void pixo_hip_free(void *p) { free_file(p); }
void pixo_hip_copy_file(void *dst, const void *src, size_t n)
{
    if (dst && src && n) big_copy(static_cast<uint8_t *>(dst), static_cast<const uint8_t *>(src), n);
}

const char *pixo_hip_last_error(void) { return t_error.c_str(); }

const char *pixo_hip_version(void) { return "pixo_hip 0.4.0 (gfx950; reference pixo 0.4.1)"; }

} // extern "C"
