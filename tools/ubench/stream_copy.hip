// stream_copy.hip — what the memory system gives a 50 MB -> 50 MB copy (the coefficient kernel's bytes: 4096x4096 RGB8 in,
// the i16 tuple out) under DIFFERENT LAUNCH SHAPES, to decide whether a persistent / LDS-DMA restructure of the kernel can
// beat its one-generation form.  No arithmetic; 7 rotating buffer pairs (700 MB > Infinity Cache);
// time per launch = HIP events over 1000 back-to-back launches / 1000, like bench.py's kernel_us.
//   A  one 16-byte element per thread (grid = 3.1 M threads / 256)
//   B  one generation like the kernel: 2048 x 192, every thread 8 x 16-byte loads first, then 8 stores (nt both ways)
//   C  persistent grid-stride: G workgroups x 256 threads, per iteration U loads then U stores (nt), G and U swept
//   D  persistent, loads by LDS-DMA (global_load_lds_dwordx4 into a per-wave ring of S slots of 1 KiB x U), stores from
//      ds_read_b128: the "persistent tiles with LDS-DMA" shape; waves per CU and ring depth swept
//   E  as D but the tile pattern of the kernel: a workgroup's unit is a 512x16-pixel tile (16 rows x 1536 B, 12288 B
//      apart) in, 24 KiB contiguous out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
constexpr size_t kBytes = 50331648, kChunks = kBytes / 16; // 3,145,728 chunks of 16 bytes

__global__ __launch_bounds__(256) void copy_a(const v4u *in, v4u *out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

__global__ __launch_bounds__(192) void copy_b(const v4u *in, v4u *out)
{
    const size_t base = (size_t)blockIdx.x * 1536 + threadIdx.x;
    v4u v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(in + base + k * 192);
#pragma unroll
    for (int k = 0; k < 8; k++) __builtin_nontemporal_store(v[k], out + base + k * 192);
}

// B with the workgroup -> chunk map permuted: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs.
//   XMAP 1: XCD x works on ONE contiguous eighth of the buffers (consecutive ids of an XCD = consecutive chunks)
//   XMAP 2: chunks dealt to the XCDs in runs of 8 (8 consecutive 24 KiB chunks per XCD, then the next XCD)
template <int XMAP>
__global__ __launch_bounds__(192) void copy_bx(const v4u *in, v4u *out)
{
    const uint32_t n = gridDim.x, id = blockIdx.x, xcd = id & 7, k = id >> 3;
    const uint32_t chunk = XMAP == 1 ? xcd * (n >> 3) + k : ((k >> 3) * 64 + xcd * 8 + (k & 7));
    const size_t base = (size_t)chunk * 1536 + threadIdx.x;
    v4u r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = __builtin_nontemporal_load(in + base + j * 192);
#pragma unroll
    for (int j = 0; j < 8; j++) __builtin_nontemporal_store(r[j], out + base + j * 192);
}

template <int U>
__global__ __launch_bounds__(256) void copy_c(const v4u *in, v4u *out, uint32_t iters)
{
    // workgroup g of G owns chunks [g * iters * 256 * U, ...): contiguous per workgroup, 4 KiB x U per iteration
    size_t at = ((size_t)blockIdx.x * iters) * (256 * U) + threadIdx.x;
    for (uint32_t it = 0; it < iters; it++, at += 256 * U) {
        v4u v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = __builtin_nontemporal_load(in + at + k * 256);
#pragma unroll
        for (int k = 0; k < U; k++) __builtin_nontemporal_store(v[k], out + at + k * 256);
    }
}

// software-pipelined: the next iteration's loads are in flight while this one's stores go out
template <int U>
__global__ __launch_bounds__(256) void copy_c2(const v4u *in, v4u *out, uint32_t iters)
{
    size_t at = ((size_t)blockIdx.x * iters) * (256 * U) + threadIdx.x;
    v4u v[U], n[U];
#pragma unroll
    for (int k = 0; k < U; k++) v[k] = __builtin_nontemporal_load(in + at + k * 256);
    for (uint32_t it = 0; it < iters; it++, at += 256 * U) {
        if (it + 1 < iters) {
#pragma unroll
            for (int k = 0; k < U; k++) n[k] = __builtin_nontemporal_load(in + at + 256 * U + k * 256);
        }
#pragma unroll
        for (int k = 0; k < U; k++) __builtin_nontemporal_store(v[k], out + at + k * 256);
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = n[k];
    }
}

// LDS-DMA: one wave-instruction moves 1 KiB (64 lanes x 16 B) from global memory to LDS at m0 + lane * 16
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// D: every wave owns a ring of S slots x U KiB in LDS; slot s is filled by U DMA instructions, read back with ds_read_b128
// and stored.  Waves are independent (no barrier): vmcnt counts the wave's own DMAs and stores in order.
template <int WAVES, int S, int U>
__global__ __launch_bounds__(64 * WAVES) void copy_d(const uint8_t *in, uint8_t *out, uint32_t iters)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t ring = (uint32_t)(uintptr_t)lds + wave * (S * U * 1024); // LDS byte address of the wave's ring
    // wave w of workgroup g owns iters consecutive units of U KiB
    const size_t unit0 = ((size_t)blockIdx.x * WAVES + wave) * iters;
    const uint8_t *src = in + unit0 * (U * 1024) + lane * 16;
    uint8_t *dst = out + unit0 * (U * 1024) + lane * 16;
    // prologue: fill S - 1 slots
#pragma unroll
    for (int s = 0; s < S - 1; s++)
        if ((uint32_t)s < iters) {
#pragma unroll
            for (int k = 0; k < U; k++) glds16(src + (size_t)s * (U * 1024) + k * 1024, ring + (s * U + k) * 1024);
        }
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t nx = it + S - 1;
        if (nx < iters) {
            const uint32_t slot = nx % S;
#pragma unroll
            for (int k = 0; k < U; k++) glds16(src + (size_t)nx * (U * 1024) + k * 1024, ring + (slot * U + k) * 1024);
            // vmcnt retires in issue order.  Issued after slot `it`'s DMAs: per later iteration one group of U DMAs and — in
            // steady state — one group of U stores: 2 (S - 1) groups may stay outstanding; during the first S - 1 iterations
            // fewer stores exist, so only the S - 1 DMA groups may
            if (it >= (uint32_t)(S - 1)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (S - 1) * U) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 1) * U) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const uint32_t slot = it % S;
        v4u v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = *(const v4u *)(lds + wave * (S * U * 1024) + (slot * U + k) * 1024 + lane * 16);
#pragma unroll
        for (int k = 0; k < U; k++) __builtin_nontemporal_store(v[k], (v4u *)(dst + (size_t)it * (U * 1024) + k * 1024));
    }
}

// E: the kernel's tile pattern.  A workgroup of 3 waves loops over tiles (512x16 px = 16 rows x 1536 B, rows 12288 B
// apart); a tile = 24 KiB = 24 DMA instructions, 8 per wave; ring of S tile slots per workgroup; output 24 KiB contiguous.
template <int S>
__global__ __launch_bounds__(192) void copy_e(const uint8_t *in, uint8_t *out, uint32_t tiles_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t ring = (uint32_t)(uintptr_t)lds;
    const uint32_t t0 = blockIdx.x * tiles_per_wg;
    auto issue = [&](uint32_t t, uint32_t slot) {
        const uint32_t tx = t & 7, ty = t >> 3;
        const uint8_t *tile = in + (size_t)ty * 16 * 12288 + tx * 1536;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int ch = (wave * 8 + k) * 64 + lane, row = ch / 96, col = (ch % 96) * 16; // 16-byte chunk of the tile
            glds16(tile + (size_t)row * 12288 + col, ring + slot * 24576 + (wave * 8 + k) * 1024);
        }
    };
#pragma unroll
    for (int s = 0; s < S - 1; s++) if ((uint32_t)s < tiles_per_wg) issue(t0 + s, s);
    for (uint32_t it = 0; it < tiles_per_wg; it++) {
        const uint32_t nx = it + S - 1;
        if (nx < tiles_per_wg) {
            issue(t0 + nx, nx % S);
            if (it >= (uint32_t)(S - 1)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (S - 1) * 8) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 1) * 8) : "memory");
        }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t slot = it % S;
        v4u v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = *(const v4u *)(lds + slot * 24576 + (wave * 8 + k) * 1024 + lane * 16);
        uint8_t *o = out + (size_t)(t0 + it) * 24576 + wave * 8192 + lane * 16;
#pragma unroll
        for (int k = 0; k < 8; k++) __builtin_nontemporal_store(v[k], (v4u *)(o + k * 1024));
    }
}

// F / G: one direction only, one generation (2048 x 192, 8 x 16 bytes per thread): what the memory system gives reads and
// writes by themselves — a copy cannot take less than both together (HBM's data bus carries one direction at a time)
__global__ __launch_bounds__(192) void read_only(const v4u *in, v4u *out)
{
    const size_t base = (size_t)blockIdx.x * 1536 + threadIdx.x;
    v4u a = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) { const v4u v = __builtin_nontemporal_load(in + base + k * 192); a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) out[base] = a; // (never: the inputs are constant bytes)
}
__global__ __launch_bounds__(192) void write_only(const v4u *in, v4u *out)
{
    const size_t base = (size_t)blockIdx.x * 1536 + threadIdx.x;
    const v4u v = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
#pragma unroll
    for (int k = 0; k < 8; k++) __builtin_nontemporal_store(v, out + base + k * 192);
}
__global__ __launch_bounds__(192) void empty_kernel(const v4u *in, v4u *out) {}

int main()
{
    const int NB = 7;
    uint8_t *in[NB], *out[NB];
    for (int i = 0; i < NB; i++) { CK(hipMalloc(&in[i], kBytes)); CK(hipMalloc(&out[i], kBytes)); CK(hipMemset(in[i], i + 1, kBytes)); CK(hipMemset(out[i], 0, kBytes)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint8_t *host = (uint8_t *)malloc(kBytes);
    auto check = [&](const char *name) { // buffer pair 0 must be an exact copy
        hipMemcpy(host, out[0], kBytes, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < kBytes; i++) bad += host[i] != 1;
        if (bad) printf("   !! %s: %zu bytes wrong\n", name, bad);
        hipMemset(out[0], 0, kBytes);
    };
    auto time = [&](const char *name, auto launch) {
        int n = 0;
        for (int i = 0; i < 3000; i++, n++) launch(in[n % NB], out[n % NB]);
        { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) printf("  sync: %s\n", hipGetErrorString(e)); e = hipGetLastError(); if (e != hipSuccess) printf("  launch: %s\n", hipGetErrorString(e)); }
        if (name[0] != '-' && name[0] != 'F') check(name);
        float best = 1e9f, sum = 0;
        const int K = 1000, R = 3;
        for (int r = 0; r < R; r++) {
            hipEventRecord(e0);
            for (int i = 0; i < K; i++, n++) launch(in[n % NB], out[n % NB]);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best; sum += ms;
        }
        const double bytes = name[0] == '-' ? 0.0 : (name[0] == 'F' || name[0] == 'G' ? 50.331648 : 100.663296);
        printf("%-86s %7.2f us (best block %6.2f)  %5.2f TB/s\n", name, sum / R * 1e3 / K, best * 1e3 / K, bytes / (sum / R * 1e3 / K));
        fflush(stdout);
    };
    time("-  empty kernel, 2048 x 192 (the launch itself)", [&](uint8_t *i, uint8_t *o) { hipLaunchKernelGGL(empty_kernel, dim3(2048), dim3(192), 0, 0, (const v4u *)i, (v4u *)o); });
    for (int i = 0; i < NB; i++) (void)hipMemset(out[i], 0, kBytes);
    time("F  READ only, 50 MB: 2048 x 192, 8 loads per thread", [&](uint8_t *i, uint8_t *o) { hipLaunchKernelGGL(read_only, dim3(2048), dim3(192), 0, 0, (const v4u *)i, (v4u *)o); });
    time("G  WRITE only, 50 MB: 2048 x 192, 8 stores per thread", [&](uint8_t *i, uint8_t *o) { hipLaunchKernelGGL(write_only, dim3(2048), dim3(192), 0, 0, (const v4u *)i, (v4u *)o); });
    time("A  one 16-byte chunk per thread, 12288 x 256", [&](uint8_t *i, uint8_t *o) { hipLaunchKernelGGL(copy_a, dim3(kChunks / 256), dim3(256), 0, 0, (const v4u *)i, (v4u *)o); });
    time("B  one generation: 2048 x 192, 8 loads then 8 stores per thread", [&](uint8_t *i, uint8_t *o) { hipLaunchKernelGGL(copy_b, dim3(2048), dim3(192), 0, 0, (const v4u *)i, (v4u *)o); });
    time("B1 like B, every XCD on ONE contiguous eighth of the buffers", [&](uint8_t *i, uint8_t *o) { hipLaunchKernelGGL(copy_bx<1>, dim3(2048), dim3(192), 0, 0, (const v4u *)i, (v4u *)o); });
    time("B2 like B, chunks dealt to the XCDs in runs of 8 (192 KiB per XCD at a time)", [&](uint8_t *i, uint8_t *o) { hipLaunchKernelGGL(copy_bx<2>, dim3(2048), dim3(192), 0, 0, (const v4u *)i, (v4u *)o); });
    time("B  one generation (again)", [&](uint8_t *i, uint8_t *o) { hipLaunchKernelGGL(copy_b, dim3(2048), dim3(192), 0, 0, (const v4u *)i, (v4u *)o); });
    if (getenv("ONLY_B")) return 0;
#define C_CASE(U, G) { char nm[128]; snprintf(nm, sizeof nm, "C  persistent grid-stride, %d x 256, %d loads then %d stores per iteration", G, U, U); \
    const uint32_t iters = kChunks / (256 * U) / G; \
    if ((size_t)iters * G * 256 * U == kChunks) time(nm, [&](uint8_t *i, uint8_t *o) { hipLaunchKernelGGL(copy_c<U>, dim3(G), dim3(256), 0, 0, (const v4u *)i, (v4u *)o, iters); }); }
    C_CASE(4, 256) C_CASE(4, 512) C_CASE(4, 1024) C_CASE(4, 2048) C_CASE(8, 512) C_CASE(8, 1024) C_CASE(2, 2048) C_CASE(1, 2048) C_CASE(1, 4096)
#define C2_CASE(U, G) { char nm[128]; snprintf(nm, sizeof nm, "C2 persistent, software pipelined (next loads in flight during stores), %d x 256, U = %d", G, U); \
    const uint32_t iters = kChunks / (256 * U) / G; \
    if ((size_t)iters * G * 256 * U == kChunks) time(nm, [&](uint8_t *i, uint8_t *o) { hipLaunchKernelGGL(copy_c2<U>, dim3(G), dim3(256), 0, 0, (const v4u *)i, (v4u *)o, iters); }); }
    C2_CASE(4, 512) C2_CASE(4, 1024) C2_CASE(4, 2048) C2_CASE(2, 2048) C2_CASE(8, 512) C2_CASE(8, 1024)
#define D_CASE(WV, S, U, G) { char nm[160]; snprintf(nm, sizeof nm, "D  LDS-DMA ring, %d x %d waves, %d slots x %d KiB per wave (%d KiB LDS per workgroup)", G, WV, S, U, WV * S * U); \
    const uint32_t iters = kBytes / (U * 1024) / ((size_t)G * WV); \
    if ((size_t)iters * G * WV * U * 1024 == kBytes) time(nm, [&](uint8_t *i, uint8_t *o) { hipLaunchKernelGGL((copy_d<WV, S, U>), dim3(G), dim3(64 * WV), WV * S * U * 1024, 0, i, o, iters); }); }
    D_CASE(4, 2, 4, 256) D_CASE(4, 3, 4, 256) D_CASE(4, 2, 4, 512) D_CASE(4, 3, 4, 512) D_CASE(4, 4, 2, 512) D_CASE(4, 2, 4, 1024) D_CASE(4, 3, 2, 1024) D_CASE(4, 4, 1, 1024)
    D_CASE(4, 2, 2, 2048) D_CASE(4, 3, 1, 2048) D_CASE(8, 2, 4, 256) D_CASE(8, 3, 2, 256) D_CASE(8, 4, 2, 256)
#define E_CASE(S, G) { char nm[160]; snprintf(nm, sizeof nm, "E  LDS-DMA tiles (512x16 px in, 24 KiB out), %d x 192, ring of %d tile slots (%d KiB)", G, S, S * 24); \
    const uint32_t per = 2048 / G; \
    time(nm, [&](uint8_t *i, uint8_t *o) { hipLaunchKernelGGL((copy_e<S>), dim3(G), dim3(192), S * 24576, 0, i, o, per); }); }
    E_CASE(2, 256) E_CASE(3, 256) E_CASE(2, 512) E_CASE(3, 512) E_CASE(2, 1024) E_CASE(1, 2048) E_CASE(1, 1024)
    time("B  one generation (again)", [&](uint8_t *i, uint8_t *o) { hipLaunchKernelGGL(copy_b, dim3(2048), dim3(192), 0, 0, (const v4u *)i, (v4u *)o); });
    time("A  one chunk per thread (again)", [&](uint8_t *i, uint8_t *o) { hipLaunchKernelGGL(copy_a, dim3(kChunks / 256), dim3(256), 0, 0, (const v4u *)i, (v4u *)o); });
    return 0;
}
