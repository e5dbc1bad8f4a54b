// valu_rates.hip — measures issue cost (cycles per wave64 instruction per SIMD) of the VALU
// opcodes the JPEG kernel is built from, on the actual gfx950 part, plus the HBM stream
// ceiling for the kernel's own access shapes (12 B/lane loads, 16 B/lane stores).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2000;
constexpr int PER_ITER = 32;

#define R4(X) X X X X
// 8 independent chains, each instruction reads its own accumulator: no cross dependencies
#define BODY2(OP) \
  OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" \
  OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"
#define BODY3(OP) \
  OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n" \
  OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9\n"
#define BODY1(OP) \
  OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" \
  OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7\n"

#define KERNEL32(NAME, ASMBODY)                                                              \
  __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed) {                \
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,   \
             a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed | 1, c = seed * 3 + 1;       \
    for (int i = 0; i < ITERS; i++)                                                          \
      asm volatile(R4(ASMBODY)                                                               \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                   : "v"(b), "v"(c));                                                        \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;             \
  }

// 64-bit register pairs for v_pk_*_f32
#define KERNEL64(NAME, ASMBODY)                                                              \
  __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed) {                \
    typedef float f2 __attribute__((ext_vector_type(2)));                                    \
    f2 a0 = {1.0f + threadIdx.x, 2.0f}, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f,       \
       a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f, b = {1.0001f, 0.9999f}, \
       c = {0.5f, 0.25f};                                                                    \
    for (int i = 0; i < ITERS; i++)                                                          \
      asm volatile(R4(ASMBODY)                                                               \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                   : "v"(b), "v"(c));                                                        \
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                            \
    out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(s.x + s.y);                        \
  }

KERNEL32(k_add_f32, BODY2("v_add_f32"))
KERNEL32(k_mul_f32, BODY2("v_mul_f32"))
KERNEL32(k_sub_f32, BODY2("v_sub_f32"))
KERNEL32(k_fma_f32, BODY3("v_fma_f32"))
KERNEL32(k_max_f32, BODY2("v_max_f32"))
KERNEL32(k_rndne_f32, BODY1("v_rndne_f32"))
KERNEL32(k_cvt_i32_f32, BODY1("v_cvt_i32_f32"))
KERNEL32(k_cvt_f32_ubyte1, BODY1("v_cvt_f32_ubyte1"))
KERNEL32(k_cvt_f32_u32, BODY1("v_cvt_f32_u32"))
KERNEL32(k_cvt_pk_i16_i32, BODY2("v_cvt_pk_i16_i32"))
KERNEL32(k_perm_b32, BODY3("v_perm_b32"))
KERNEL32(k_pk_mad_u16, BODY3("v_pk_mad_u16"))
KERNEL32(k_pk_add_u16, BODY2("v_pk_add_u16"))
KERNEL32(k_pk_mul_lo_u16, BODY2("v_pk_mul_lo_u16"))
KERNEL32(k_pk_lshrrev_b16, BODY2("v_pk_lshrrev_b16"))
KERNEL32(k_pk_min_u16, BODY2("v_pk_min_u16"))
KERNEL32(k_and_b32, BODY2("v_and_b32"))
KERNEL32(k_add_u32, BODY2("v_add_u32"))
KERNEL32(k_lshrrev_b32, BODY2("v_lshrrev_b32"))
KERNEL32(k_mov_b32, BODY1("v_mov_b32"))
KERNEL32(k_mad_u32_u24, BODY3("v_mad_u32_u24"))
KERNEL32(k_mad_i32_i24, BODY3("v_mad_i32_i24"))
KERNEL32(k_dot4_u32_u8, BODY3("v_dot4_u32_u8"))
KERNEL32(k_mul_lo_u32, BODY2("v_mul_lo_u32"))
KERNEL32(k_lshl_add_u32, BODY3("v_lshl_add_u32"))
KERNEL32(k_add3_u32, BODY3("v_add3_u32"))
KERNEL32(k_bfe_u32, BODY3("v_bfe_u32"))
KERNEL32(k_bfi_b32, BODY3("v_bfi_b32"))
KERNEL32(k_med3_f32, BODY3("v_med3_f32"))
KERNEL32(k_mad_u16, BODY3("v_mad_u16"))
KERNEL32(k_mad_legacy_u16, BODY3("v_mad_legacy_u16"))
KERNEL32(k_cmp_ge_f32, "v_cmp_ge_f32 vcc, %0, %8\n v_cmp_ge_f32 vcc, %1, %8\n v_cmp_ge_f32 vcc, %2, %8\n v_cmp_ge_f32 vcc, %3, %8\n v_cmp_ge_f32 vcc, %4, %8\n v_cmp_ge_f32 vcc, %5, %8\n v_cmp_ge_f32 vcc, %6, %8\n v_cmp_ge_f32 vcc, %7, %8\n")
KERNEL32(k_cndmask_b32, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
KERNEL64(k_pk_add_f32, BODY2("v_pk_add_f32"))
KERNEL64(k_pk_mul_f32, BODY2("v_pk_mul_f32"))
KERNEL64(k_pk_fma_f32, BODY3("v_pk_fma_f32"))

struct Entry { const char *name; void (*fn)(unsigned *, unsigned); };
#define E(n) {#n, n}
static Entry entries[] = {
  E(k_add_f32), E(k_mul_f32), E(k_sub_f32), E(k_fma_f32), E(k_max_f32), E(k_med3_f32), E(k_pk_add_f32), E(k_pk_mul_f32), E(k_pk_fma_f32),
  E(k_rndne_f32), E(k_cvt_i32_f32), E(k_cvt_f32_ubyte1), E(k_cvt_f32_u32), E(k_cvt_pk_i16_i32), E(k_cmp_ge_f32), E(k_cndmask_b32),
  E(k_perm_b32), E(k_pk_mad_u16), E(k_pk_add_u16), E(k_pk_mul_lo_u16), E(k_pk_lshrrev_b16), E(k_pk_min_u16), E(k_mad_u16), E(k_mad_legacy_u16),
  E(k_and_b32), E(k_add_u32), E(k_lshrrev_b32), E(k_mov_b32), E(k_mad_u32_u24), E(k_mad_i32_i24), E(k_dot4_u32_u8), E(k_mul_lo_u32),
  E(k_lshl_add_u32), E(k_add3_u32), E(k_bfe_u32), E(k_bfi_b32),
};

// ---- memory stream shapes ----------------------------------------------------------------
struct u3 { unsigned a, b, c; };
__global__ __launch_bounds__(256) void k_stream_12B_16B(const unsigned *in, uint4 *out, size_t n12, size_t n16)
{ // read 12 B/lane (dwordx3), write 16 B/lane, same byte volume each way
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (; i < n12; i += stride) {
    const unsigned *p = in + i * 3;
    unsigned a = p[0], b = p[1], c = p[2];
    if (i * 3 / 4 < n16 && (i & 3) != 3) out[(i >> 2) * 3 + (i & 3)] = make_uint4(a, b, c, a ^ b);
  }
}
__global__ __launch_bounds__(256) void k_copy16(const uint4 *in, uint4 *out, size_t n)
{
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_read16(const uint4 *in, unsigned *out, size_t n)
{
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  unsigned acc = 0;
  for (; i < n; i += stride) { uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s  CUs %d  clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
  unsigned *out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4 + 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // reference: cycles for v_add_f32 is unknown too, so report ns per wave-instruction per SIMD
  printf("%-22s %10s %14s %16s\n", "opcode", "ms", "ns/inst/SIMD", "cyc@2.4GHz/inst");
  for (int wpe : {4, 1}) {
    printf("-- %d wave(s) per SIMD (blocks per CU = %d)\n", wpe, wpe);
    for (auto &en : entries) {
      dim3 grid(cus * wpe);
      hipLaunchKernelGGL(en.fn, grid, dim3(256), 0, 0, out, 12345u);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < 3; r++) hipLaunchKernelGGL(en.fn, grid, dim3(256), 0, 0, out, 12345u + r);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      double inst_per_simd = (double)ITERS * PER_ITER * wpe; // each SIMD hosts wpe waves
      double ns = ms * 1e6 / inst_per_simd;
      printf("%-22s %10.4f %14.4f %16.3f\n", en.name + 2, ms, ns, ns * 2.4);
    }
  }
  // memory streams: 1 GiB working set > 256 MiB Infinity Cache
  size_t bytes = (size_t)768 << 20;
  void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  for (int blocks_per_cu : {4, 8, 16}) {
    dim3 grid(cus * blocks_per_cu);
    auto timeit = [&](const char *name, auto launch, double moved) {
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
      printf("%-28s blocks/CU %2d  %8.3f ms  %8.1f GB/s\n", name, blocks_per_cu, ms, moved / ms / 1e6);
    };
    timeit("copy 16B/lane r+w", [&] { hipLaunchKernelGGL(k_copy16, grid, dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, bytes / 16); }, 2.0 * bytes);
    timeit("read 16B/lane", [&] { hipLaunchKernelGGL(k_read16, grid, dim3(256), 0, 0, (const uint4 *)a, out, bytes / 16); }, 1.0 * bytes);
    timeit("read 12B/lane write 16B/lane", [&] { hipLaunchKernelGGL(k_stream_12B_16B, grid, dim3(256), 0, 0, (const unsigned *)a, (uint4 *)b, bytes / 12, bytes / 16); }, 2.0 * bytes);
  }
  return 0;
}
