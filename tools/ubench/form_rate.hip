// form_rate.hip — issue cost of specific instruction FORMS (operand kinds, modifiers, encodings)
// at 2 and 4 waves per SIMD: ns and cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2000;

#define FORM(NAME, ASM)                                                                        \
    __global__ __launch_bounds__(256) void NAME(float *out, float seed, unsigned long long *clk) \
    {                                                                                          \
        float a[8], d[8];                                                                      \
        float b = seed * 0.5f, sc = seed;                                                      \
        for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x + i; d[i] = 0; }               \
        const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();       \
        for (int it = 0; it < ITERS; it++) {                                                   \
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "=v"(d[i]) : "v"(a[i]), "v"(b), "s"(sc)); \
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "=v"(a[i]) : "v"(d[i]), "v"(b), "s"(sc)); \
        }                                                                                      \
        const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();       \
        float s = 0;                                                                           \
        for (int i = 0; i < 8; i++) s += a[i] + d[i];                                          \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                               \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }       \
    }

FORM(f_add_vv, "v_add_f32 %0, %1, %2")
FORM(f_add_lit, "v_add_f32 %0, 0x4b400000, %1")
FORM(f_add_inline, "v_add_f32 %0, -0.5, %1")
FORM(f_add_sgpr, "v_add_f32 %0, %3, %1")
FORM(f_add_e64_abs, "v_add_f32_e64 %0, |%1|, %2")
FORM(f_sub_e64_neg, "v_add_f32_e64 %0, -%1, %2")
FORM(f_mul_lit, "v_mul_f32 %0, 0x3f3504f3, %1")
FORM(f_mul_sgpr, "v_mul_f32 %0, %3, %1")
FORM(f_fma_vvv, "v_fma_f32 %0, %1, %2, %2")
FORM(f_fma_abs, "v_fma_f32 %0, |%1|, %2, |%2|")
FORM(f_fma_sgpr, "v_fma_f32 %0, %1, %3, %2")
FORM(f_fma_abs_sgpr, "v_fma_f32 %0, |%1|, %3, |%2|")
FORM(f_fmac, "v_fmac_f32 %0, %1, %2")
FORM(f_fmaak, "v_fmaak_f32 %0, %1, %2, 0x3f3504f3")
FORM(f_fmamk, "v_fmamk_f32 %0, %1, 0x35000000, %2")
FORM(f_and, "v_and_b32 %0, %1, %2")
FORM(f_and_lit, "v_and_b32 %0, 0x7fffffff, %1")
FORM(f_or_sdwa, "v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
FORM(f_cvt_ubyte1, "v_cvt_f32_ubyte1 %0, %1")
FORM(f_bitop3, "v_bitop3_b32 %0, %1, %2, %2 bitop3:0x80")
FORM(f_and_or, "v_and_or_b32 %0, %1, 63, %2")
FORM(f_perm, "v_perm_b32 %0, %1, %2, %3")
FORM(f_max, "v_max_f32 %0, %1, %2")
FORM(f_min_u32, "v_min_u32 %0, %1, %2")
FORM(f_mul_u24, "v_mul_u32_u24 %0, %1, %2")
FORM(f_xor, "v_xor_b32 %0, %1, %2")
FORM(f_sub_u32, "v_sub_u32 %0, %1, %2")
FORM(f_lshl_lit, "v_lshlrev_b32 %0, 8, %1")
FORM(f_alignbit, "v_alignbit_b32 %0, %1, %2, 8")
FORM(f_pk_add_u16, "v_pk_add_u16 %0, %1, %2")
FORM(f_add_u16, "v_add_u16 %0, %1, %2")
FORM(f_mul_lo_u16, "v_mul_lo_u16 %0, %1, %2")
FORM(f_cvt_f32_i32, "v_cvt_f32_i32 %0, %1")
FORM(f_rndne, "v_rndne_f32 %0, %1")
FORM(f_dot4c, "v_dot4c_i32_i8 %0, %1, %2")
FORM(f_dot2c_f16, "v_dot2c_f32_f16 %0, %1, %2")
FORM(f_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
FORM(f_add_dpp, "v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")

// forms on 64-bit operands (register pairs): packed f32, f64 min / max — the "float" interface is kept, the pairs are built
// from two of the floats
#define FORM64(NAME, ASM)                                                                      \
    __global__ __launch_bounds__(256) void NAME(float *out, float seed, unsigned long long *clk) \
    {                                                                                          \
        double a[8], d[8];                                                                     \
        double b = seed * 0.5; const double sc = seed;                                         \
        for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x + i; d[i] = 0; }               \
        const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();       \
        for (int it = 0; it < ITERS; it++) {                                                   \
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "=v"(d[i]) : "v"(a[i]), "v"(b), "s"(sc)); \
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "=v"(a[i]) : "v"(d[i]), "v"(b), "s"(sc)); \
        }                                                                                      \
        const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();       \
        double s = 0;                                                                          \
        for (int i = 0; i < 8; i++) s += a[i] + d[i];                                          \
        out[blockIdx.x * 256 + threadIdx.x] = (float)s;                                        \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }       \
    }
FORM64(g_pk_fma_vvv, "v_pk_fma_f32 %0, %1, %2, %2")
FORM64(g_pk_fma_bcast_sgpr, "v_pk_fma_f32 %0, %1, %3, %2 op_sel_hi:[0,1,0]")
FORM64(g_pk_add_f32, "v_pk_add_f32 %0, %1, %2")
FORM64(g_pk_mul_f32, "v_pk_mul_f32 %0, %1, %2")
FORM64(g_min_f64, "v_min_f64 %0, %1, %2")
FORM64(g_max_f64, "v_max_f64 %0, %1, %2")
FORM64(g_add_f64, "v_add_f64 %0, %1, %2")
FORM(f_lerp_u8, "v_lerp_u8 %0, %1, %2, %2")
FORM(f_sad_u8, "v_sad_u8 %0, %1, %2, %2")
FORM(f_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %2")
FORM(f_dot4_u32_u8, "v_dot4_u32_u8 %0, %1, %2, %2")
FORM(f_pk_min_u16, "v_pk_min_u16 %0, %1, %2")
FORM(f_pk_ashr_i16, "v_pk_ashrrev_i16 %0, 15, %1")
FORM(f_pk_sub_i16, "v_pk_sub_i16 %0, %1, %2")
FORM(f_bfe_u32, "v_bfe_u32 %0, %1, 6, 4")
FORM(f_lshl_or, "v_lshl_or_b32 %0, %1, 8, %2")
FORM(f_lshl_add, "v_lshl_add_u32 %0, %1, 1, %2")
FORM(f_add3, "v_add3_u32 %0, %1, %2, %2")
FORM(f_or3, "v_or3_b32 %0, %1, %2, %2")
FORM(f_bfi, "v_bfi_b32 %0, %1, %2, %2")
FORM(f_cmp_ne_u32, "v_cmp_ne_u32 vcc, %1, %2")
FORM(f_cmp_lt_f32, "v_cmp_lt_f32 vcc, %1, %2")
FORM(f_cndmask, "v_cndmask_b32 %0, %1, %2, vcc")
FORM(f_alignbyte, "v_alignbyte_b32 %0, %1, %2, 1")
FORM(f_cvt_ubyte0, "v_cvt_f32_ubyte0 %0, %1")
FORM(f_mul_lo_u32, "v_mul_lo_u32 %0, %1, %2")
FORM(f_readlane_free, "v_mov_b32 %0, %1")

struct Entry { const char *name; void (*fn)(float *, float, unsigned long long *); };
#define E(n) {#n, n}
static Entry entries[] = {E(f_add_vv), E(f_add_lit), E(f_add_inline), E(f_add_sgpr), E(f_add_e64_abs), E(f_sub_e64_neg), E(f_mul_lit), E(f_mul_sgpr),
    E(f_fma_vvv), E(f_fma_abs), E(f_fma_sgpr), E(f_fma_abs_sgpr), E(f_fmac), E(f_fmaak), E(f_fmamk), E(f_and), E(f_and_lit), E(f_or_sdwa), E(f_cvt_ubyte1),
    E(f_bitop3), E(f_and_or), E(f_perm), E(f_max), E(f_min_u32), E(f_mul_u24), E(f_xor), E(f_sub_u32), E(f_lshl_lit), E(f_alignbit), E(f_pk_add_u16),
    E(f_add_u16), E(f_mul_lo_u16), E(f_cvt_f32_i32), E(f_rndne), E(f_dot4c), E(f_dot2c_f16), E(f_mov_dpp), E(f_add_dpp),
    E(g_pk_fma_vvv), E(g_pk_fma_bcast_sgpr), E(g_pk_add_f32), E(g_pk_mul_f32), E(g_min_f64), E(g_max_f64), E(g_add_f64), E(f_lerp_u8), E(f_sad_u8),
    E(f_mad_u32_u24), E(f_dot4_u32_u8), E(f_pk_min_u16), E(f_pk_ashr_i16), E(f_pk_sub_i16), E(f_bfe_u32), E(f_lshl_or), E(f_lshl_add), E(f_add3), E(f_or3),
    E(f_bfi), E(f_cmp_ne_u32), E(f_cmp_lt_f32), E(f_cndmask), E(f_alignbyte), E(f_cvt_ubyte0), E(f_mul_lo_u32), E(f_readlane_free)};

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    unsigned long long *clk; CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-18s %s\n", "form", "cycles per wave-instruction per SIMD at 2 / 4 waves per SIMD (loop overhead ~0.1 included)");
    for (auto &e : entries) {
        printf("%-18s", e.name);
        for (int w = 2; w <= 4; w += 2) {
            dim3 grid(cus * w);
            hipLaunchKernelGGL(e.fn, grid, dim3(256), 0, 0, out, 1.0f, clk);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(e.fn, grid, dim3(256), 0, 0, out, 1.0f + r, clk);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
            unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double ghz = (double)h[0] / ((double)h[1] * 10.0);
            printf("  %6.2f", ms * 1e6 / ((double)ITERS * 16 * w) * ghz);
        }
        printf("\n");
    }
    return 0;
}
