// jpeg_api.cpp — the extern "C" JPEG entry points declared in include/pixo_hip.h.  No CPU fallback exists: without a
// usable GPU every compute entry point fails with PIXO_ERR_COMPRESSION and says so.
#include <algorithm>

#include "dispatch_gate.hpp"
#include "capi_internal.hpp"

using namespace pixo_capi;

namespace {
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
    if (!o.progressive && debug().host_entropy) { // (experiments: the host twin of the entropy stage)
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
    if (o.progressive) {
        HIP_TRY(hipMemcpyAsync(c.d_px, data, px_bytes, hipMemcpyHostToDevice, c.stream));
        uint8_t *direct = nullptr; // pinned / registered caller storage: the scans are copied from the device straight into it
        if (dest && dest_cap > 1) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, dest) == hipSuccess && at.type == hipMemoryTypeHost) direct = dest;
            else (void)hipGetLastError(); // (plain malloc'd memory is "invalid value" to the runtime: not an error)
        }
        return progressive_to_view(c.d_px, o, g, c, spill, file, file_len, direct, dest_cap);
    }
    // baseline: the entropy stage launches the uploads and the coefficient kernel itself — band by band for images of
    // 2048x2048 pixels and more, so that bands are transformed and coded while the next ones cross PCIe and the file's
    // first pieces travel back meanwhile (pieces.cpp)
    int16_t *dy, *dcb, *dcr;
    if ((rc = coeffs_reserve(c, g, &dy, &dcb, &dcr))) return rc;
    PixelSource src{c.d_px, &o, &g, dy, dcb, dcr, data};
    return device_entropy_to_pinned(c, dy, dcb, dcr, o, g, c.stream, file, file_len, 1, nullptr, nullptr, dest, dest_cap, own_malloc, &src);
}

int fail_tuple_trellis()
{ // trellis quantisation happens between the transform and the tuple (src/jpeg/mod.rs:932-976): a tuple entry cannot apply it
    return fail(PIXO_ERR_COMPRESSION, "Compression error: trellis_quant needs the pixels: quantise the tuple with the trellis "
                                      "quantiser first and clear the flag, or use an entry point that takes pixels");
}
} // namespace

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
    // (the library's copy threads and a huge-page hint for large planes: 50 MB into a caller's fresh arrays)
    big_copy(reinterpret_cast<uint8_t *>(y), reinterpret_cast<const uint8_t *>(hy), g.y_blocks * 128);
    if (g.c_blocks) {
        big_copy(reinterpret_cast<uint8_t *>(cb), reinterpret_cast<const uint8_t *>(hcb), g.c_blocks * 128);
        big_copy(reinterpret_cast<uint8_t *>(cr), reinterpret_cast<const uint8_t *>(hcr), g.c_blocks * 128);
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
        uint8_t *direct = nullptr; // pinned / registered storage: the scans are copied from the device straight into it
        if (output && capacity) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, output) == hipSuccess && at.type == hipMemoryTypeHost) direct = output;
            else (void)hipGetLastError(); // (plain malloc'd memory is "invalid value" to the runtime: not an error)
        }
        if ((rc = progressive_to_view(d_pixels, *options, g, *c, spill, &file, &n, direct, capacity))) return rc;
        *out_len = n;
        if (n > capacity) return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(n) + " bytes");
        if (file != output) big_copy(output, file, n);
        return PIXO_OK;
    }
    if (debug().host_entropy) { // assembled on the host: copy if it fits
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

} // extern "C"

namespace {
// The entropy-coded bytes of a batch in c.e_out: one coefficient launch + one pass of the entropy stage (the images are
// segments of the single-pass kernels).  Only for option sets that allow it (see the callers).
int batch_on_device(Context &c, const void *d_pixels, const pixo_jpeg_options &o, const pixo_host::Geometry &g, uint32_t batch,
                    std::vector<uint8_t> &head, std::vector<uint64_t> &starts, bool *gaps)
{ // *gaps: the scans lie in c.e_out with room for EOI + the next file's headers between them (the files' final spacing)
    std::vector<uint8_t> probe_head;
    pixo_host::file_headers(probe_head, o, pixo_host::HuffSet::standard()); // (a one-pass batch has the standard tables)
    const uint32_t gap = static_cast<uint32_t>(probe_head.size() + 2);
    const float *qt_all = nullptr;
    int rc = device_tables(c.device, &qt_all);
    if (rc) return rc;
    const size_t coef_bytes = (g.y_blocks + 2 * g.c_blocks) * 128 * batch;
    if ((rc = c.reserve_coef(coef_bytes))) return rc;
    int16_t *dy = static_cast<int16_t *>(c.d_coef), *dcb = dy + g.y_blocks * 64 * batch, *dcr = dcb + g.c_blocks * 64 * batch;
    // (round 6: the entropy stage gets the PIXELS — an RGB batch goes through the fused pixel -> scan kernel, every image a segment,
    // and never writes the tuple; otherwise the stage launches the coefficient kernel over the batch itself)
    const PixelSource src{d_pixels, &o, &g, dy, dcb, dcr};
    const uint8_t *unused = nullptr;
    size_t scan_bytes = 0;
    return device_entropy_to_pinned(c, dy, dcb, dcr, o, g, c.stream, &unused, &scan_bytes, batch, &starts, nullptr, nullptr, 0, nullptr, &src, &head,
                                    gap, gaps);
}
bool batch_in_one_pass(const pixo_jpeg_options &o, const pixo_host::Geometry &g, uint32_t batch, size_t px_bytes)
{
    return batch > 1 && !o.progressive && !o.optimize_huffman && !scan_has_restart_markers(o, g) && px_bytes % 4 == 0;
}
} // namespace

extern "C" {

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
    PIXO_REQUIRE(d_pixels);
    Context *c = nullptr;
    if ((rc = context_on_current_device(&c))) return rc;
    const pixo_jpeg_options &o = *options;
    const pixo_host::Geometry g = pixo_host::geometry(o.width, o.height, o.color_type, o.subsampling);
    const size_t px_bytes = static_cast<size_t>(o.width) * o.height * (g.gray ? 1 : 3);
    for (uint32_t i = 0; i < batch; ++i) { files[i] = nullptr; lens[i] = 0; }
    auto release = [&](int code) { for (uint32_t i = 0; i < batch; ++i) { pixo_hip_free(files[i]); files[i] = nullptr; } return code; };
    if (!batch_in_one_pass(o, g, batch, px_bytes)) { // per-image tables or segments inside the images: one image at a time
        for (uint32_t i = 0; i < batch; ++i)
            if ((rc = pixo_hip_jpeg_encode_device(static_cast<const uint8_t *>(d_pixels) + i * px_bytes, options, &files[i], &lens[i]))) return release(rc);
        return PIXO_OK;
    }
    std::vector<uint8_t> head;
    std::vector<uint64_t> starts;
    bool gaps = false;
    if ((rc = batch_on_device(*c, d_pixels, o, g, batch, head, starts, &gaps))) return rc;
    const size_t hdr = head.size(), scan_bytes = static_cast<size_t>(starts[batch]), gap = gaps ? hdr + 2 : 0;
    for (uint32_t i = 0; i < batch; ++i) lens[i] = hdr + static_cast<size_t>(starts[i + 1] - starts[i]) - (i + 1 < batch ? gap : 0) + 2;
    // Round 6: every file's block comes from the library's pool of PINNED host memory (pieces.cpp pool_take: resident pages that
    // pixo_hip_free gives back) and its entropy-coded bytes are copied from the device straight into it — 64 x 1080p noise: 26.8 ->
    // ~2.5 ms a batch, where fresh malloc'd blocks cost 22,000 page faults.  The pool exhausted (or debug switch plain_host): the
    // old way below.
    bool pooled = true;
    for (uint32_t i = 0; i < batch && pooled; ++i)
        if (!(files[i] = pool_take(lens[i]))) pooled = false;
    if (pooled) {
        hipError_t e = hipSuccess;
        for (uint32_t i = 0; i < batch && e == hipSuccess; ++i) {
            const size_t seg = lens[i] - hdr - 2;
            if (seg) e = hipMemcpyAsync(files[i] + hdr, c->e_out.as<uint8_t>() + starts[i], seg, hipMemcpyDeviceToHost, c->stream);
        }
        for (uint32_t i = 0; i < batch; ++i) { // (headers and EOI while the copies run: they touch other bytes)
            std::memcpy(files[i], head.data(), hdr);
            files[i][lens[i] - 2] = 0xFF; files[i][lens[i] - 1] = 0xD9;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            for (uint32_t i = 0; i < batch; ++i) { pixo_hip_free(files[i]); files[i] = nullptr; }
            return hip_fail(e, "device-to-host copy of the batch files");
        }
        return PIXO_OK;
    }
    for (uint32_t i = 0; i < batch; ++i) { if (files[i]) pixo_hip_free(files[i]); files[i] = nullptr; }
    // the stuffed bytes cross PCIe once, into the context's pinned buffer (a device-to-host copy into fresh pageable blocks
    // would make the runtime pin new pages every call); from there into the files the caller will own — fresh memory,
    // page-fault bound: several threads (see big_copy)
    if ((rc = c->reserve_hfile(scan_bytes ? scan_bytes : 1))) return rc;
    HIP_TRY(hipMemcpyAsync(c->h_file, c->e_out.p, scan_bytes, hipMemcpyDeviceToHost, c->stream));
    for (uint32_t i = 0; i < batch; ++i) {
        files[i] = static_cast<uint8_t *>(std::malloc(lens[i]));
        if (!files[i]) { (void)hipStreamSynchronize(c->stream); return release(fail(PIXO_ERR_COMPRESSION, "Compression error: out of host memory")); }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t total = scan_bytes + static_cast<size_t>(batch) * (hdr + 2);
    const unsigned t = static_cast<unsigned>(std::min<size_t>(std::min<size_t>(debug().copy_threads, batch), total >> 21));
    run_on_threads(t ? t : 1, [&](unsigned k) {
        for (uint32_t i = k; i < batch; i += (t ? t : 1)) {
            const size_t seg = lens[i] - hdr - 2;
            uint8_t *p = files[i];
            std::memcpy(p, head.data(), hdr);
            std::memcpy(p + hdr, c->h_file + starts[i], seg);
            p[hdr + seg] = 0xFF; p[hdr + seg + 1] = 0xD9;
        }
    });
    return PIXO_OK;
}

int pixo_hip_jpeg_encode_batch_device_into(const void *d_pixels, const pixo_jpeg_options *options, uint32_t batch,
                                           uint8_t *arena, size_t capacity, size_t *offsets, size_t *lens)
{
    PIXO_REQUIRE(options);
    PIXO_REQUIRE(offsets);
    PIXO_REQUIRE(lens);
    std::string msg;
    int rc = pixo_host::validate(*options, false, 0, msg);
    if (rc) return fail(rc, msg);
    if (batch == 0 || batch > 65535) return fail(PIXO_ERR_COMPRESSION, "Compression error: batch must be 1..65535");
    PIXO_REQUIRE(d_pixels);
    if (capacity && !arena) return fail(PIXO_ERR_COMPRESSION, "Compression error: null argument 'arena'");
    Context *c = nullptr;
    if ((rc = context_on_current_device(&c))) return rc;
    const pixo_jpeg_options &o = *options;
    const pixo_host::Geometry g = pixo_host::geometry(o.width, o.height, o.color_type, o.subsampling);
    const size_t px_bytes = static_cast<size_t>(o.width) * o.height * (g.gray ? 1 : 3);
    for (uint32_t i = 0; i < batch; ++i) { offsets[i] = 0; lens[i] = 0; }
    bool arena_pinned = false, arena_device = false;
    if (arena) {
        hipPointerAttribute_t pa;
        if (hipPointerGetAttributes(&pa, arena) == hipSuccess) {
            arena_pinned = pa.type == hipMemoryTypeHost;
            arena_device = pa.type == hipMemoryTypeDevice;
        } else (void)hipGetLastError(); // (plain malloc'd memory is "invalid value" to the runtime: not an error)
    }
    if (!batch_in_one_pass(o, g, batch, px_bytes) && arena_device) { // per-image tables / segments inside the images, files to stay in HBM:
        size_t at = 0;                                                // each image through a host file, then host-to-device behind the one before
        hipError_t e = hipSuccess;
        for (uint32_t i = 0; i < batch && e == hipSuccess; ++i) {
            uint8_t *f = nullptr;
            size_t n = 0;
            if ((rc = pixo_hip_jpeg_encode_device(static_cast<const uint8_t *>(d_pixels) + i * px_bytes, options, &f, &n))) return rc;
            offsets[i] = at; lens[i] = n;
            if (at + n <= capacity) e = hipMemcpy(arena + at, f, n, hipMemcpyHostToDevice);
            at += n;
            pixo_hip_free(f);
        }
        if (e != hipSuccess) return hip_fail(e, "host-to-device copy of a batch file");
        if (at > capacity) return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(at) + " bytes");
        return PIXO_OK;
    }
    if (!batch_in_one_pass(o, g, batch, px_bytes)) { // one image at a time, each straight into its place behind the one before
        size_t at = 0;
        bool fits = true;
        for (uint32_t i = 0; i < batch; ++i) {
            size_t n = 0;
            static uint8_t nowhere;
            uint8_t *dst = fits && arena && at < capacity ? arena + at : &nowhere;
            rc = pixo_hip_jpeg_encode_device_into(static_cast<const uint8_t *>(d_pixels) + i * px_bytes, options, dst, dst == &nowhere ? 0 : capacity - at, &n);
            if (rc == PIXO_ERR_BUFFER_TOO_SMALL) fits = false;
            else if (rc) return rc;
            offsets[i] = at; lens[i] = n;
            at += n;
        }
        if (!fits) return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(at) + " bytes");
        return PIXO_OK;
    }
    // Sub-batches alternate between two contexts (two streams, two sets of buffers): the device-to-host copy of one
    // sub-batch's files runs while the next one's kernels do — 64 x 1080p noise: 88.9 MB over PCIe are 1.7 ms, the kernels
    // of the whole batch 0.4 ms; in one pass they added up (2.08 ms).  A sub-batch's place in the arena is known when the
    // one before has been sized (its entropy pass ends with that read-back), before its bytes have moved.
    // Only where the files are large enough for their copy to matter: smooth content (0.6 bytes per block) is 3.6 MB for
    // the same batch, and eight passes cost 0.66 ms where one takes 0.43.  The context remembers the last batch's bytes per
    // block; an unknown or changed content is found out after the first sub-batch, the rest then goes in one pass.
    const size_t blocks_per_image = g.y_blocks + 2 * g.c_blocks;
    // Round 5: in between (photograph-like content, 3-8 bytes per block: 64 x 1080p = 22 MB) four sub-batches — 0.76 -> 0.61 ms,
    // where eight take 0.73 (profiles/r05_batch_parts.txt).
    constexpr uint32_t kWorthIt = 8, kMedium = 3; // bytes per block
    auto parts_for = [&](uint32_t one_plus_per_block) -> uint32_t {
        if (px_bytes * batch < (size_t{64} << 20)) return 1;
        if (one_plus_per_block == 0 || one_plus_per_block > kWorthIt) return std::min<uint32_t>(std::max<uint32_t>(batch / 8, 1), 8);
        if (one_plus_per_block > kMedium) return std::min<uint32_t>(std::max<uint32_t>(batch / 16, 1), 4);
        return 1;
    };
    uint32_t parts = parts_for(c->batch_per_block);
    if (debug().batch_parts) parts = std::min<uint32_t>(debug().batch_parts, batch);
    Context *second = nullptr;
    if (parts > 1) {
        second = pool().take(c->device);
        if (second && (second->ensure() || order_after_producer(*second))) { pool().give(second); second = nullptr; }
    }
    std::vector<uint8_t> head;
    size_t at = 0;
    uint32_t first = 0;
    rc = PIXO_OK;
    if (arena && !arena_pinned && !arena_device) advise_huge(arena, capacity);
    for (uint32_t part = 0; part < parts && !rc; ++part) {
        uint32_t nb = (batch - first + (parts - part) - 1) / (parts - part);
        Context &cx = (second && (part & 1)) ? *second : *c;
        std::vector<uint64_t> starts;
        bool gaps = false;
        if ((rc = batch_on_device(cx, static_cast<const uint8_t *>(d_pixels) + static_cast<size_t>(first) * px_bytes, o, g, nb, head, starts, &gaps))) break;
        const size_t hdr = head.size(), gap = gaps ? hdr + 2 : 0;
        const size_t at0 = at;
        for (uint32_t i = 0; i < nb; ++i) {
            offsets[first + i] = at;
            lens[first + i] = hdr + static_cast<size_t>(starts[i + 1] - starts[i]) - (i + 1 < nb ? gap : 0) + 2;
            at += lens[first + i];
        }
        if (at <= capacity) {
            hipError_t e = hipSuccess;
            if (gaps && arena_device) { // the files stay in HBM (a caller that gathers them over RCCL, pixo_amd/sharded.py): one device-to-device copy
                const size_t run = static_cast<size_t>(starts[nb]);
                if (run) e = hipMemcpyAsync(arena + at0 + hdr, cx.e_out.p, run, hipMemcpyDeviceToDevice, cx.stream);
            } else if (gaps && !arena_pinned) { // pageable arena: through the context's pinned buffer + the copy threads (a copy straight
                                         // into pageable pages makes the runtime fault them in and pin them as it goes)
                const size_t run = static_cast<size_t>(starts[nb]);
                if (run) {
                    if ((rc = cx.reserve_hfile(run))) break;
                    e = hipMemcpyAsync(cx.h_file, cx.e_out.p, run, hipMemcpyDeviceToHost, cx.stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(cx.stream);
                    if (e == hipSuccess) big_copy(arena + at0 + hdr, cx.h_file, run);
                }
            } else if (gaps) { // the scans lie in the device buffer at their files' final spacing: ONE copy, the host fills the gaps in afterwards
                const size_t run = static_cast<size_t>(starts[nb]);
                if (run) e = hipMemcpyAsync(arena + at0 + hdr, cx.e_out.p, run, hipMemcpyDeviceToHost, cx.stream);
            } else { // (multi-pass kernels: every file's entropy-coded bytes by a copy of its own)
                for (uint32_t i = 0; i < nb && e == hipSuccess; ++i) {
                    const size_t seg = lens[first + i] - hdr - 2;
                    if (seg) e = hipMemcpyAsync(arena + offsets[first + i] + hdr, cx.e_out.as<uint8_t>() + starts[i], seg,
                                                arena_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, cx.stream);
                }
            }
            if (e != hipSuccess) rc = hip_fail(e, "device-to-host copy of the batch");
        }
        first += nb;
        const size_t per_block = (at - at0) / (static_cast<size_t>(nb) * blocks_per_image);
        c->batch_per_block = static_cast<uint32_t>(1 + per_block);
        if (part == 0 && parts > 1 && !debug().batch_parts) { // (what the content really is: the rest in as many passes as that is worth)
            const uint32_t want = parts_for(c->batch_per_block);
            if (want < parts) parts = std::max<uint32_t>(want, 2);
        } // (small files after all: everything else in one more pass)
    }
    { // (both streams: also after an error, the second context goes back to the pool idle)
        hipError_t e = hipStreamSynchronize(c->stream);
        if (second) {
            const hipError_t e2 = hipStreamSynchronize(second->stream);
            if (e == hipSuccess) e = e2;
            pool().give(second);
        }
        if (!rc && e != hipSuccess) rc = hip_fail(e, "device-to-host copy of the batch");
    }
    if (rc) return rc;
    if (at > capacity) return fail(PIXO_ERR_BUFFER_TOO_SMALL, "output buffer too small: need " + std::to_string(at) + " bytes");
    const size_t hdr = head.size();
    if (arena_device) { // headers and EOI markers: one small upload (offsets + the header bytes) and one launch, a workgroup per seam
        std::vector<uint64_t> meta(batch + 1 + (hdr + 7) / 8);
        for (uint32_t i = 0; i < batch; ++i) meta[i] = offsets[i];
        meta[batch] = at;
        std::memcpy(meta.data() + batch + 1, head.data(), hdr);
        hipError_t e = c->e_seams.reserve(meta.size() * 8);
        if (e == hipSuccess) e = hipMemcpyAsync(c->e_seams.p, meta.data(), meta.size() * 8, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = pixo_dev::launch_batch_seams(arena, c->e_seams.as<unsigned long long>(), batch, static_cast<uint32_t>(hdr), c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return hip_fail(e, "headers of the batch files");
        return PIXO_OK;
    }
    for (uint32_t i = 0; i < batch; ++i) {
        uint8_t *p = arena + offsets[i];
        std::memcpy(p, head.data(), hdr);
        p[lens[i] - 2] = 0xFF; p[lens[i] - 1] = 0xD9;
    }
    return PIXO_OK;
}

uint64_t pixo_hip_debug_lookback_fallbacks(void) { return lookback_fallbacks(); }
int pixo_hip_debug_dispatch_gate(uint64_t *waits, uint64_t *timeouts)
{
    unsigned long long w = 0, t = 0;
    pixo_dev::dispatch_gate_stats(&w, &t);
    if (waits) *waits = w;
    if (timeouts) *timeouts = t;
    return PIXO_OK;
}

} // extern "C"
