// stream_copy.hip — a MEASUREMENT entry, not part of the encode path: what the memory system gives a plain copy of the
// coefficient kernel's bytes in the coefficient kernel's launch shape (bench.py times it in the same run, on the same
// box and clocks, right behind the metric: roofline.copy_us_same_run).
//
// Shape (tools/ubench/stream_copy.hip, shape B — the fastest of all shapes measured there, profiles/r04_stream_copy_ceiling.txt):
// one generation of 192-thread workgroups like jpeg_coeffs_kernel's, every workgroup moves 24 KiB — the bytes of one
// 512x16-pixel tile —: each thread issues 8 non-temporal 16-byte loads, then 8 non-temporal 16-byte stores.  No arithmetic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pixo_hip.h"
#include "capi_internal.hpp"

namespace {
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
constexpr size_t kChunk = 192 * 8 * 16; // bytes per workgroup

__global__ __launch_bounds__(192) void stream_copy_kernel(const v4u *in, v4u *out)
{
    const size_t base = (size_t)blockIdx.x * 1536 + threadIdx.x;
    v4u v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(in + base + k * 192);
#pragma unroll
    for (int k = 0; k < 8; k++) __builtin_nontemporal_store(v[k], out + base + k * 192);
}
} // namespace

extern "C" int pixo_hip_debug_stream_copy(const void *d_in, void *d_out, size_t bytes, void *stream)
{
    if (!d_in || !d_out || bytes == 0 || bytes % kChunk != 0 || bytes / kChunk > 0x7FFFFFFFu ||
        (reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) % 16 != 0)
        return pixo_capi::fail(PIXO_ERR_COMPRESSION, "Compression error: pixo_hip_debug_stream_copy wants 16-byte aligned device pointers and a multiple of 24576 bytes");
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)(bytes / kChunk)), dim3(192), 0, static_cast<hipStream_t>(stream),
                       static_cast<const v4u *>(d_in), static_cast<v4u *>(d_out));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pixo_capi::hip_fail(e, "stream_copy_kernel");
    return PIXO_OK;
}
