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
constexpr size_t kPiece = 192 * 16;     // bytes one instruction of a workgroup moves

__global__ __launch_bounds__(192) void stream_copy_kernel(const v4u *in, v4u *out)
{
    const size_t base = (size_t)blockIdx.x * 1536 + threadIdx.x;
    v4u v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(in + base + k * 192);
#pragma unroll
    for (int k = 0; k < 8; k++) __builtin_nontemporal_store(v[k], out + base + k * 192);
}

// The same shape for a kernel that writes more or less than it reads (4:4:4: 12 KiB of pixels in, 24 KiB of coefficients out per
// workgroup; the fused pixel -> scan kernel: 24 KiB in, a few KiB out): every workgroup issues R loads and W stores of 16 bytes
// per thread; what is stored is the loaded data folded together (every byte read is used, no byte is invented).
template <int R, int W> __global__ __launch_bounds__(192) void stream_io_kernel(const v4u *in, v4u *out)
{
    const size_t ib = (size_t)blockIdx.x * (192 * R) + threadIdx.x, ob = (size_t)blockIdx.x * (192 * W) + threadIdx.x;
    v4u v[R];
#pragma unroll
    for (int k = 0; k < R; k++) v[k] = __builtin_nontemporal_load(in + ib + k * 192);
#pragma unroll
    for (int k = 0; k < W; k++) {
        v4u o = v[k % R];
        if (W < R) { // (fewer stores than loads: fold the loads that have no store of their own into this one)
#pragma unroll
            for (int j = k + W; j < R; j += W) o ^= v[j];
        }
        __builtin_nontemporal_store(o, out + ob + k * 192);
    }
}
template <int R> bool launch_io(int w, unsigned wgs, const v4u *in, v4u *out, hipStream_t s)
{
    switch (w) {
    case 1: hipLaunchKernelGGL((stream_io_kernel<R, 1>), dim3(wgs), dim3(192), 0, s, in, out); return true;
    case 2: hipLaunchKernelGGL((stream_io_kernel<R, 2>), dim3(wgs), dim3(192), 0, s, in, out); return true;
    case 4: hipLaunchKernelGGL((stream_io_kernel<R, 4>), dim3(wgs), dim3(192), 0, s, in, out); return true;
    case 8: hipLaunchKernelGGL((stream_io_kernel<R, 8>), dim3(wgs), dim3(192), 0, s, in, out); return true;
    case 16: hipLaunchKernelGGL((stream_io_kernel<R, 16>), dim3(wgs), dim3(192), 0, s, in, out); return true;
    default: return false;
    }
}
// The engine clock while every SIMD issues vector instructions (four wavefronts each): shader-clock ticks (s_memtime) over
// constant-clock ticks (s_memrealtime, 100 MHz) across ~0.5 ms of dependent v_add_f32 chains, read by one thread.  The chip clocks
// down under vector load (2.0-2.4 GHz measured, tools/ubench/memtime.hip): the issue roofline's denominator is THIS, not the peak.
__global__ __launch_bounds__(256) void engine_clock_kernel(unsigned long long *out, float *sink, int iters)
{
    float a0 = (float)threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.0f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++)
        asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == -1.0f) sink[0] = a0; // (never true: keeps the chains alive)
}
} // namespace

extern "C" int pixo_hip_debug_engine_clock(void *stream, double *hz)
{
    if (!hz) return pixo_capi::fail(PIXO_ERR_COMPRESSION, "Compression error: null argument 'hz'");
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned long long *d = nullptr, h[2] = {0, 0};
    hipError_t e = hipMalloc(&d, 64);
    if (e != hipSuccess) return pixo_capi::hip_fail(e, "hipMalloc");
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    for (int rep = 0; rep < 2 && e == hipSuccess; rep++) { // (the second launch runs at the clocks the first one settled)
        hipLaunchKernelGGL(engine_clock_kernel, dim3((unsigned)cus * 4), dim3(256), 0, s, d, reinterpret_cast<float *>(d + 4), 20000);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) return pixo_capi::hip_fail(e, "engine_clock_kernel");
    if (h[1] == 0) return pixo_capi::fail(PIXO_ERR_COMPRESSION, "Compression error: the constant clock did not advance");
    *hz = (double)h[0] / (double)h[1] * 100e6;
    return PIXO_OK;
}

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

extern "C" int pixo_hip_debug_stream_io(const void *d_in, void *d_out, uint32_t workgroups, uint32_t loads, uint32_t stores, void *stream)
{
    auto pow2 = [](uint32_t v) { return v >= 1 && v <= 16 && (v & (v - 1)) == 0; };
    if (!d_in || !d_out || workgroups == 0 || workgroups > 0x7FFFFFFFu || !pow2(loads) || !pow2(stores) ||
        (reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) % 16 != 0)
        return pixo_capi::fail(PIXO_ERR_COMPRESSION, "Compression error: pixo_hip_debug_stream_io wants 16-byte aligned device pointers, loads and stores in {1, 2, 4, 8, 16}");
    const v4u *in = static_cast<const v4u *>(d_in);
    v4u *out = static_cast<v4u *>(d_out);
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool ok = false;
    switch (loads) {
    case 1: ok = launch_io<1>((int)stores, workgroups, in, out, s); break;
    case 2: ok = launch_io<2>((int)stores, workgroups, in, out, s); break;
    case 4: ok = launch_io<4>((int)stores, workgroups, in, out, s); break;
    case 8: ok = launch_io<8>((int)stores, workgroups, in, out, s); break;
    case 16: ok = launch_io<16>((int)stores, workgroups, in, out, s); break;
    default: break;
    }
    if (!ok) return pixo_capi::fail(PIXO_ERR_COMPRESSION, "Compression error: pixo_hip_debug_stream_io: unsupported shape");
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pixo_capi::hip_fail(e, "stream_io_kernel");
    (void)kPiece;
    return PIXO_OK;
}
