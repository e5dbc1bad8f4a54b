// bands.cpp — one image across several GPUs: per-band entropy coding + bit-exact splice.
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>

#include "capi_internal.hpp"

using namespace pixo_capi;

extern "C" {

struct pixo_hip_band_encoder {
    pixo_jpeg_options image{}, band{};   // the whole image / the same options with the band's height
    pixo_host::Geometry g{};             // of the band
    uint32_t parts = 1, index = 0, rows = 0, row_begin = 0;
    Context *c = nullptr;                // adopted from the pool for the encoder's lifetime
    int16_t *dy = nullptr, *dcb = nullptr, *dcr = nullptr;
    ScanJob job;
    int stage = 0;                       // 0 created, 1 coefficients done, 2 lengths done
    int16_t last_dc[3] = {0, 0, 0};
    int16_t prev_dc[3] = {0, 0, 0};      // what `lengths` was called with (a pack that has to start over needs them again)
    std::vector<uint64_t> counts;        // ... and the statistics of all bands (optimised tables), empty = none
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
    { const int rc_t = c.ensure_totals(); if (rc_t) return rc_t; }
    int16_t *h = reinterpret_cast<int16_t *>(c.h_totals + Context::kTotalsWords - 1); // (the mailbox's last word: no kernel of the entropy stage writes it)
    HIP_TRY(pixo_dev::launch_last_dcs(e->dy, e->g.y_blocks, e->dcb, e->dcr, e->g.c_blocks, h, c.stream));
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
    for (int i = 0; i < 3; ++i) e->prev_dc[i] = prev_dc[i];
    e->counts.clear();
    if (total_counts) e->counts.assign(total_counts, total_counts + PIXO_HIP_COUNT_WORDS);
    int rc = scan_begin(c, e->job, e->dy, e->dcb, e->dcr, e->band, e->g, 1, prev_dc);
    if (rc) return rc;
    rc = scan_lengths(c, e->job, e->image, e->g, c.stream, total_counts);
    if (rc == kRetryMultipass) { // a single-pass kernel gave up waiting: the band again with the multi-pass kernels
        RetryMultipass scope;
        if ((rc = scan_begin(c, e->job, e->dy, e->dcb, e->dcr, e->band, e->g, 1, prev_dc))) return rc;
        rc = scan_lengths(c, e->job, e->image, e->g, c.stream, total_counts);
    }
    if (rc) return rc == kRetryMultipass ? fail(PIXO_ERR_COMPRESSION, "Compression error: the entropy kernels could not make progress") : rc;
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
    if (rc == kRetryMultipass) { // (the stuffing kernel gave up waiting: lengths and packing again, multi-pass)
        RetryMultipass scope;
        const uint64_t *tc = e->counts.empty() ? nullptr : e->counts.data();
        if ((rc = scan_begin(c, e->job, e->dy, e->dcb, e->dcr, e->band, e->g, 1, e->prev_dc))) return rc;
        if ((rc = scan_lengths(c, e->job, e->image, e->g, c.stream, tc))) return rc;
        rc = scan_pack(c, e->job, c.stream, bit_offset, &head, &tail_bits, &tail);
    }
    if (rc) return rc == kRetryMultipass ? fail(PIXO_ERR_COMPRESSION, "Compression error: the entropy kernels could not make progress") : rc;
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
    if (e->parts <= 2) big_copy(dst, c.h_file, body); // (few bands: nobody else is copying — the library's copy threads)
    else std::memcpy(dst, c.h_file, body);
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

// Band workers: threads that live as long as the process, one per band slot, each with the library context of its own
// (bound to the device of the band it last served; re-bound when a call lists another device for that slot).  A call of
// pixo_hip_jpeg_encode_multi hands every worker the same closure and waits for all of them: no thread is created and no
// context taken from the pool per call (round 2 spawned n threads and created n encoders every time).  One multi-device
// encode runs at a time per process (the workers are shared); callers on other threads wait their turn.
class BandWorkers {
  public:
    // runs body(k) for k in [0, n) on workers 0 .. n - 1 and returns when all have finished; false = the threads could not
    // be created (nothing has run)
    bool run(unsigned n, const std::function<void(unsigned)> &body)
    {
        std::lock_guard<std::mutex> turn(turn_);
        try {
            while (workers_.size() < n) {
                std::unique_ptr<Worker> w(new Worker);
                Worker *raw = w.get();
                w->thread = std::thread([this, raw] { loop(*raw); });
                workers_.push_back(std::move(w));
            }
        } catch (...) { // (std::system_error: no more threads; std::bad_alloc) — the workers that exist stay idle
            return false;
        }
        {
            std::lock_guard<std::mutex> lock(m_);
            body_ = &body;
            pending_ = n;
            for (unsigned k = 0; k < n; ++k) workers_[k]->job = static_cast<int>(k);
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lock(m_);
        done_.wait(lock, [&] { return pending_ == 0; });
        body_ = nullptr;
        return true;
    }
    // body(k) on every worker that exists (pixo_hip_trim: their thread-local device buffers); nothing when there are none
    void run_on_existing(const std::function<void(unsigned)> &body)
    {
        unsigned n = 0;
        {
            std::lock_guard<std::mutex> turn(turn_);
            n = static_cast<unsigned>(workers_.size());
        }
        if (n) (void)run(n, body);
    }
  private:
    struct Worker { std::thread thread; int job = -1; };
    void loop(Worker &w)
    {
        for (;;) {
            const std::function<void(unsigned)> *body = nullptr;
            int job = -1;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_.wait(lock, [&] { return w.job >= 0; });
                job = w.job;
                w.job = -1;
                body = body_;
            }
            (*body)(static_cast<unsigned>(job));
            {
                std::lock_guard<std::mutex> lock(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::mutex turn_, m_;
    std::condition_variable cv_, done_;
    std::vector<std::unique_ptr<Worker>> workers_; // (never destroyed: the threads are detached from process exit like the context pool)
    const std::function<void(unsigned)> *body_ = nullptr;
    unsigned pending_ = 0;
};
extern "C++" BandWorkers &band_workers_instance() // (C++ linkage: this file's body sits inside extern "C")
{
    static BandWorkers *w = new BandWorkers; // leaked on purpose: worker threads must not be joined from a static destructor
    return *w;
}
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
            if (!r && !(file = alloc_file(layout.file_len))) r = fail(PIXO_ERR_COMPRESSION, "Compression error: out of host memory");
            step(r);
        }
        barrier.arrive();
        // every band's bytes go straight to their final place, over its own GPU's PCIe link, on its own thread
        if (!failed.load()) step(pixo_hip_band_encoder_copy_body(b.enc, file + layout.body_off[k]));
        pixo_hip_band_encoder_destroy(b.enc);
        b.enc = nullptr;
    };
    if (!band_workers_instance().run(parts, body)) return fail(PIXO_ERR_COMPRESSION, "Compression error: could not start the band worker threads");
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


// ---- a batch of images over the GPUs of one process (SURVEY §8e "C3 batch"; round 5) --------------------------------------
namespace {
// Device storage of a band worker between calls (grow-only; the worker threads live as long as the process): the images it
// received from another GPU, and its files before they travel.
struct BatchWorkerBuffers {
    int device = -1;
    void *px = nullptr, *arena = nullptr;
    size_t px_cap = 0, arena_cap = 0;
    void drop()
    {
        if (device >= 0) {
            DeviceScope on(device);
            if (px) (void)hipFree(px);
            if (arena) (void)hipFree(arena);
        }
        px = arena = nullptr; px_cap = arena_cap = 0;
    }
    hipError_t reserve(void **p, size_t *cap, size_t want)
    {
        if (*cap >= want) return hipSuccess;
        if (*p) (void)hipFree(*p);
        *p = nullptr; *cap = 0;
        const hipError_t e = hipMalloc(p, want);
        if (e == hipSuccess) *cap = want;
        return e;
    }
};
thread_local BatchWorkerBuffers t_batch_buffers;
} // namespace
// pixo_hip_trim: the workers' device buffers go back to the driver (each worker frees its own, on its own thread)
extern "C++" {
namespace pixo_capi {
void drop_batch_worker_buffers()
{
    band_workers_instance().run_on_existing([](unsigned) { t_batch_buffers.drop(); t_batch_buffers.device = -1; });
}
} // namespace pixo_capi
}

int pixo_hip_jpeg_encode_batch_multi(const void *pixels, const pixo_jpeg_options *options, uint32_t batch, const int *devices, uint32_t n_devices,
                                     uint8_t *arena, size_t capacity, size_t *offsets, size_t *lens)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(devices);
    PIXO_REQUIRE(offsets);
    PIXO_REQUIRE(lens);
    const pixo_jpeg_options &o = *options;
    std::string msg;
    int rc = pixo_host::validate(o, false, 0, msg);
    if (rc) return fail(rc, msg);
    PIXO_REQUIRE(pixels);
    if (batch == 0) return fail(PIXO_ERR_COMPRESSION, "Compression error: empty batch");
    if (n_devices == 0 || n_devices > 1024) return fail(PIXO_ERR_COMPRESSION, "Compression error: need 1..1024 devices");
    if (arena == nullptr && capacity != 0) return fail(PIXO_ERR_COMPRESSION, "Compression error: null arena with a capacity");
    const pixo_host::Geometry g = pixo_host::geometry(o.width, o.height, o.color_type, o.subsampling);
    const size_t image_bytes = static_cast<size_t>(o.width) * o.height * (g.gray ? 1 : 3);
    // where the pixels are: in some GPU's memory (images then reach the other GPUs by peer copies, one peer per xGMI link)
    // or in host memory (every GPU fetches its own images over its own PCIe link)
    int src_device = -1;
    {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, pixels) == hipSuccess && at.type == hipMemoryTypeDevice) src_device = at.device;
        else (void)hipGetLastError(); // (plain host memory is "invalid value" to the runtime: not an error)
    }
    // Device pixels may still be being written on the caller's producer stream (pixo_hip_set_producer_stream; default: the NULL
    // stream).  The work below runs on worker threads with streams of their own: the caller's stream is drained first.
    if (src_device >= 0) {
        DeviceScope on(src_device);
        if (on.err != hipSuccess) return hip_fail(on.err, "hipSetDevice");
        HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(pixo_hip_get_producer_stream())));
    }
    const uint32_t parts = n_devices;
    struct Share {
        uint32_t first = 0, count = 0;
        std::vector<size_t> offs, lens;
        size_t bytes = 0; // this share's files, back to back
        int rc = PIXO_OK;
        std::string error;
    };
    std::vector<Share> shares(parts);
    for (uint32_t k = 0; k < parts; ++k) { // contiguous runs of images, as even as the batch allows
        shares[k].first = static_cast<uint32_t>(static_cast<uint64_t>(batch) * k / parts);
        shares[k].count = static_cast<uint32_t>(static_cast<uint64_t>(batch) * (k + 1) / parts) - shares[k].first;
    }
    PhaseBarrier barrier(parts);
    std::atomic<bool> failed{false};
    std::atomic<bool> too_small{false};
    size_t total = 0;
    auto body = [&](unsigned k) {
        Share &sh = shares[k];
        auto step = [&](int r) { if (r && !sh.rc) { sh.rc = r; sh.error = t_error; failed.store(true); } };
        auto hip_step = [&](hipError_t e, const char *what) { if (e != hipSuccess) step(hip_fail(e, what)); };
        BatchWorkerBuffers &buf = t_batch_buffers;
        const int dev = devices[k];
        if (sh.count) {
            step(pixo_hip_set_device(dev)); // (binds this worker's library context and the HIP device of this thread)
            if (!sh.rc) hip_step(hipSetDevice(dev), "hipSetDevice");
            if (buf.device != dev) { buf.drop(); buf.device = dev; }
            // ---- the share's images into this GPU's memory
            const uint8_t *src = static_cast<const uint8_t *>(pixels) + static_cast<size_t>(sh.first) * image_bytes;
            const void *local = src;
            const size_t px_bytes = static_cast<size_t>(sh.count) * image_bytes;
            if (!sh.rc && src_device != dev) {
                hip_step(buf.reserve(&buf.px, &buf.px_cap, px_bytes), "hipMalloc (images of a batch share)");
                if (!sh.rc) {
                    if (src_device >= 0) hip_step(hipMemcpyPeer(buf.px, dev, src, src_device, px_bytes), "hipMemcpyPeer (images of a batch share)");
                    else hip_step(hipMemcpy(buf.px, src, px_bytes, hipMemcpyHostToDevice), "upload of a batch share");
                    local = buf.px;
                }
            }
            sh.offs.assign(sh.count, 0); sh.lens.assign(sh.count, 0);
            if (parts == 1 && !sh.rc) { // one GPU: nothing to place among other shares — the files go straight into the caller's arena, their
                // way over PCIe overlapping the kernels of the next sub-batch (no device arena, no second pass over the bytes)
                const int r = pixo_hip_jpeg_encode_batch_device_into(local, options, sh.count, arena, capacity, sh.offs.data(), sh.lens.data());
                for (uint32_t i = 0; i < sh.count; ++i) { offsets[i] = sh.offs[i]; lens[i] = sh.lens[i]; }
                if (r == PIXO_ERR_BUFFER_TOO_SMALL) { total = sh.offs[sh.count - 1] + sh.lens[sh.count - 1]; too_small.store(true); }
                else step(r);
                return;
            }
            // ---- its files, complete with headers and EOI, back to back in this GPU's memory (grow and repeat when the guess
            // was short: pixo_hip_jpeg_encode_batch_device_into fills in the lengths either way)
            size_t want = std::max<size_t>(px_bytes / 3, size_t{1} << 16);
            for (int attempt = 0; !sh.rc && attempt < 3; ++attempt) {
                hip_step(buf.reserve(&buf.arena, &buf.arena_cap, want), "hipMalloc (files of a batch share)");
                if (sh.rc) break;
                const int r = pixo_hip_jpeg_encode_batch_device_into(local, options, sh.count, static_cast<uint8_t *>(buf.arena), buf.arena_cap,
                                                                     sh.offs.data(), sh.lens.data());
                if (r == PIXO_ERR_BUFFER_TOO_SMALL && attempt < 2) { want = sh.offs[sh.count - 1] + sh.lens[sh.count - 1] + 4096; continue; }
                if (r == PIXO_ERR_BUFFER_TOO_SMALL) // (the sizes changed between three identical calls: not the CALLER's arena that is too small)
                    step(fail(PIXO_ERR_COMPRESSION, "Compression error: a batch share's size changed between identical calls"));
                else step(r);
                break;
            }
            if (!sh.rc) sh.bytes = sh.offs[sh.count - 1] + sh.lens[sh.count - 1];
        }
        barrier.arrive(); // ---- exchange: every share's file lengths -> where every file goes in the caller's arena
        if (k == 0 && !failed.load()) {
            size_t at = 0;
            for (uint32_t j = 0; j < parts; ++j)
                for (uint32_t i = 0; i < shares[j].count; ++i) {
                    offsets[shares[j].first + i] = at;
                    lens[shares[j].first + i] = shares[j].lens[i];
                    at += shares[j].lens[i];
                }
            total = at;
            if (total > capacity) too_small.store(true);
        }
        barrier.arrive();
        // ---- every share's run of files to its final place, over its own GPU's PCIe link, on its own thread
        if (!failed.load() && !too_small.load() && sh.count && sh.bytes)
            hip_step(hipMemcpy(arena + offsets[sh.first], buf.arena, sh.bytes, hipMemcpyDeviceToHost), "device-to-host copy of a batch share's files");
    };
    if (!band_workers_instance().run(parts, body)) return fail(PIXO_ERR_COMPRESSION, "Compression error: could not start the worker threads");
    for (Share &sh : shares)
        if (sh.rc) { const int r = sh.rc; const std::string e = sh.error; return fail(r, e); }
    if (too_small.load()) return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(total) + " bytes");
    return PIXO_OK;
}

} // extern "C"
