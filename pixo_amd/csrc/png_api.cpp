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
    HIP_TRY(hipMemcpyAsync(c.p_in.p, data, in_bytes, hipMemcpyHostToDevice, c.stream));
    if ((rc = png_filter_on_device(c, c.p_in.p, width, height, bytes_per_pixel, run, seq, c.p_out.p, adler32))) return rc;
    // The stream's way to the caller (round 3).  A device-to-host copy straight into pageable storage makes the runtime
    // fault in and pin the pages as it goes: 30-39 ms for the 67 MB of a 4096x4096 RGBA image, where the bytes need 1.3 ms
    // on the link.  So, like the JPEG files: into the context's pinned buffer in pieces of 8 MiB, each piece copied on by
    // the library's copy threads while the next one crosses PCIe, with a huge-page hint for storage not yet touched.
    if (out_bytes < (size_t{4} << 20)) {
        HIP_TRY(hipMemcpy(out, c.p_out.p, out_bytes, hipMemcpyDeviceToHost));
        return PIXO_OK;
    }
    if ((rc = c.reserve_hfile(out_bytes))) return rc;
    advise_huge(out, out_bytes);
    constexpr size_t kPiece = size_t{8} << 20;
    const size_t pieces = (out_bytes + kPiece - 1) / kPiece;
    std::vector<hipEvent_t> arrived(pieces, nullptr);
    hipError_t e = hipSuccess;
    for (size_t i = 0; i < pieces && e == hipSuccess; ++i) {
        const size_t off = i * kPiece, n = std::min(kPiece, out_bytes - off);
        e = hipEventCreateWithFlags(&arrived[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipMemcpyAsync(c.h_file + off, static_cast<const uint8_t *>(c.p_out.p) + off, n, hipMemcpyDeviceToHost, c.stream);
        if (e == hipSuccess) e = hipEventRecord(arrived[i], c.stream);
    }
    for (size_t i = 0; i < pieces && e == hipSuccess; ++i) {
        const size_t off = i * kPiece, n = std::min(kPiece, out_bytes - off);
        e = hipEventSynchronize(arrived[i]);
        if (e == hipSuccess) big_copy(out + off, c.h_file + off, n);
    }
    if (e != hipSuccess) (void)hipStreamSynchronize(c.stream); // (nothing of ours in flight when the events go)
    for (hipEvent_t ev : arrived)
        if (ev) (void)hipEventDestroy(ev);
    if (e != hipSuccess) return hip_fail(e, "device-to-host copy of the filtered stream");
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

} // extern "C"
