// png_api.cpp — the extern "C" PNG row-filter entry points (config 5: apply_filters + the zlib wrapper's Adler-32).
#include "capi_internal.hpp"
#include "png_filter.hpp"

#include <algorithm>
#include <vector>

using namespace pixo_capi;

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

// A large image from host pixels to caller storage, band by band (round 3).  Rows are independent once the row above is on
// the device, so: the calling thread uploads bands of ~8 MiB back to back on the upload stream (a pageable source blocks it
// for the copy's duration anyway); each band's kernel and the download of its filtered rows into the context's PINNED
// buffer follow on the context's stream as soon as the band has arrived; and while the next band uploads, the band before
// last is copied on into the caller's storage by the library's copy threads (huge-page hint first: the result array of
// a call is usually fresh).  A device-to-host copy straight into that pageable storage made the runtime fault in and pin
// its pages as it went: 30-39 ms for the 67 MB of a 4096x4096 RGBA image whose bytes need 1.3 ms each way on the link.
// The stateful AdaptiveFast (row 0 decides for all) runs as one launch after the whole upload.
int png_filter_in_bands(Context &c, const uint8_t *data, uint32_t width, uint32_t height, uint32_t bpp, int run, bool seq,
                        uint8_t *out, uint32_t *adler)
{
    const size_t row_in = static_cast<size_t>(width) * bpp, row_out = row_in + 1;
    const size_t out_bytes = row_out * height;
    int rc = c.reserve_hfile(out_bytes);
    if (rc) { // (the pinned staging is an optimisation — a 16384x16384 RGBA image wants 1.3 GB of it.  Without it:
              // the whole image in one go, its rows straight into the caller's storage, as before round 3)
        (void)hipGetLastError();
        HIP_TRY(hipMemcpyAsync(c.p_in.p, data, row_in * height, hipMemcpyHostToDevice, c.stream));
        if ((rc = png_filter_on_device(c, c.p_in.p, width, height, bpp, run, seq, c.p_out.p, adler))) return rc;
        HIP_TRY(hipMemcpy(out, c.p_out.p, out_bytes, hipMemcpyDeviceToHost));
        return PIXO_OK;
    }
    HIP_TRY(c.p_sums.reserve(static_cast<size_t>(height) * 16));
    HIP_TRY(c.p_scratch.reserve(16));
    if (static_cast<size_t>(height) * 16 > c.hsums_cap) {
        if (c.h_sums) (void)hipHostFree(c.h_sums);
        c.h_sums = nullptr; c.hsums_cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c.h_sums), static_cast<size_t>(height) * 16, hipHostMallocDefault));
        c.hsums_cap = static_cast<size_t>(height) * 16;
    }
    if (!c.upload_stream) HIP_TRY(hipStreamCreateWithFlags(&c.upload_stream, hipStreamNonBlocking));
    advise_huge(out, out_bytes);
    const uint32_t band_rows = seq ? height : static_cast<uint32_t>(std::max<size_t>(1, (size_t{8} << 20) / row_in));
    const uint32_t bands = (height + band_rows - 1) / band_rows;
    std::vector<hipEvent_t> up(bands, nullptr), down(bands, nullptr);
    hipError_t e = hipSuccess;
    auto copy_on = [&](uint32_t b) { // band b's filtered rows: pinned buffer -> the caller's storage
        e = hipEventSynchronize(down[b]);
        const size_t off = static_cast<size_t>(b) * band_rows * row_out;
        const size_t n = std::min(static_cast<size_t>(band_rows) * row_out, out_bytes - off);
        if (e == hipSuccess) big_copy(out + off, c.h_file + off, n);
    };
    for (uint32_t b = 0; b < bands && e == hipSuccess; ++b) {
        const uint32_t r0 = b * band_rows, rows = std::min(band_rows, height - r0);
        e = hipEventCreateWithFlags(&up[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&down[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipMemcpyAsync(static_cast<uint8_t *>(c.p_in.p) + r0 * row_in, data + r0 * row_in, rows * row_in, hipMemcpyHostToDevice, c.upload_stream);
        if (e == hipSuccess) e = hipEventRecord(up[b], c.upload_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c.stream, up[b], 0);
        if (e == hipSuccess) e = pixo_dev::launch_png_filter_rows(c.p_in.p, width, height, bpp, run, seq, c.p_out.p, c.p_sums.as<unsigned long long>(),
                                                                  c.p_scratch.as<int>(), r0, rows, c.stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c.h_file + r0 * row_out, static_cast<const uint8_t *>(c.p_out.p) + r0 * row_out, rows * row_out, hipMemcpyDeviceToHost, c.stream);
        if (e == hipSuccess) e = hipEventRecord(down[b], c.stream);
        if (e == hipSuccess && b >= 2) copy_on(b - 2);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(c.h_sums, c.p_sums.p, static_cast<size_t>(height) * 16, hipMemcpyDeviceToHost, c.stream);
    for (uint32_t b = bands >= 2 ? bands - 2 : 0; b < bands && e == hipSuccess; ++b) copy_on(b);
    const hipError_t e2 = hipStreamSynchronize(c.stream); // (also after an error: nothing of ours in flight when the events go)
    (void)hipStreamSynchronize(c.upload_stream);
    if (e == hipSuccess) e = e2;
    for (hipEvent_t ev : up) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : down) if (ev) (void)hipEventDestroy(ev);
    if (e != hipSuccess) return hip_fail(e, "PNG filter stage in bands");
    *adler = combine_adler(c.h_sums, height, row_out);
    return PIXO_OK;
}
} // namespace

extern "C" {

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
    if (out_bytes < (size_t{4} << 20)) { // small: one copy each way
        HIP_TRY(hipMemcpyAsync(c.p_in.p, data, in_bytes, hipMemcpyHostToDevice, c.stream));
        if ((rc = png_filter_on_device(c, c.p_in.p, width, height, bytes_per_pixel, run, seq, c.p_out.p, adler32))) return rc;
        HIP_TRY(hipMemcpy(out, c.p_out.p, out_bytes, hipMemcpyDeviceToHost));
        return PIXO_OK;
    }
    return png_filter_in_bands(c, data, width, height, bytes_per_pixel, run, seq, out, adler32);
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

} // extern "C"
