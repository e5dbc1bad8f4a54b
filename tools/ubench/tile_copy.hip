// tile_copy.hip — what the coefficient kernel's MEMORY ACCESS PATTERN costs by itself, and what other
// patterns would cost: 2048 workgroups x 192 threads move one 4096x4096 RGB8 image (50 MB in) to a
// coefficient-tuple-shaped output (50 MB out, 4:2:0 layout) with no arithmetic.
//   read  R0: tile 128 x 64 px  (64 rows x 384 B, 12288 B apart)      R1: tile 512 x 16 px (16 rows x 1536 B)
//   write W0: 64-byte half blocks (4 lanes x 16 B per block, the two halves ~1 us apart)
//         W1: whole 128-byte blocks (8 lanes x 16 B per block: 1 KiB contiguous per wave store)
// Output segments follow the tile: R0 -> Y 4 x 4 KiB + chroma 8 x 1 KiB, R1 -> Y 16 KiB + chroma 2 x 4 KiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr size_t kPitch = 12288, kYBytes = 33554432, kCBytes = 8388608;
struct u32x3 { uint32_t a, b, c; };
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int R, int W, int SPIN>
__global__ __launch_bounds__(192) void tile_copy(const uint8_t *in, uint8_t *out)
{
    __shared__ uint32_t lds[64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t id = blockIdx.x;
    const uint32_t tx = R == 0 ? id % 32 : id % 8, ty = R == 0 ? id / 32 : id / 8; // (R == 2: as R == 1)
    const uint8_t *base = in + (R == 0 ? (size_t)ty * 64 * kPitch + tx * 384 : (size_t)ty * 16 * kPitch + tx * 1536);
    uint32_t acc = 0;
    if (R == 2) { // the 512x16 tile with 16-byte loads: 16 rows x 96 lanes x 16 B = 1536 chunks
#pragma unroll
        for (int pass = 0; pass < 8; pass++) {
            const int idx = pass * 192 + tid, row = idx / 96, col = (idx % 96) * 16;
            const uint4 v = *reinterpret_cast<const uint4 *>(in + (size_t)(id / 8) * 16 * kPitch + (id % 8) * 1536 + (size_t)row * kPitch + col);
            acc ^= v.x + v.y + v.z + v.w;
        }
    } else
#pragma unroll
    for (int pass = 0; pass < 11; pass++) {
        const int idx = pass * 192 + tid;
        const int row = R == 0 ? idx >> 5 : idx >> 7, col = (R == 0 ? idx & 31 : idx & 127) * 12;
        if (idx < 2048) {
            const u32x3 v = *reinterpret_cast<const u32x3 *>(base + (size_t)row * kPitch + col);
            acc ^= v.a + v.b + v.c;
        }
    }
    if (lane == 0) lds[wave] = acc;
    __syncthreads();
    acc ^= lds[(wave + 1) % 3];
    // wave-local block index -> address
    auto blk = [&](int b) -> uint8_t * {
        if (wave < 2) { // 64 luminance blocks
            const int g = wave * 64 + b; // 0..127 in tile order
            if (R == 0) { const int m = g >> 5, k = g & 31; return out + ((size_t)((ty * 4 + m) * 256 + tx * 8) * 4 + k) * 128; }
            return out + ((size_t)(ty * 256 + tx * 32) * 4 + g) * 128;
        }
        const int plane = b >> 5, k = b & 31; // 32 Cb + 32 Cr
        uint8_t *p = out + kYBytes + plane * kCBytes;
        if (R == 0) { const int m = k >> 3, j = k & 7; return p + ((size_t)(ty * 4 + m) * 256 + tx * 8 + j) * 128; }
        return p + ((size_t)ty * 256 + tx * 32 + k) * 128;
    };
    const v4u v = {acc, acc + 1, acc + 2, acc + 3};
    if (W == 0) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(blk(j * 16 + (lane >> 2)) + half * 64 + (lane & 3) * 16));
            if (half == 0 && SPIN) { // the quantiser's second half sits between the two stores
                uint32_t x = acc;
                for (int i = 0; i < SPIN; i++) asm volatile("v_add_u32 %0, %0, %0" : "+v"(x));
                if (x == 0x12345) lds[0] = x;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++)
            __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(blk(j * 8 + (lane >> 3)) + (lane & 7) * 16));
    }
}

int main()
{
    const int NB = 7;
    uint8_t *in[NB], *out[NB];
    for (int i = 0; i < NB; i++) { CK(hipMalloc(&in[i], 50331648)); CK(hipMalloc(&out[i], 50331648)); CK(hipMemset(in[i], i + 1, 50331648)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto kernel) {
        int n = 0;
        for (int i = 0; i < 5000; i++, n++) { hipLaunchKernelGGL(kernel, dim3(2048), dim3(192), 0, 0, in[n % NB], out[n % NB]); }
        { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) printf("  sync: %s\n", hipGetErrorString(e)); e = hipGetLastError(); if (e != hipSuccess) printf("  launch: %s\n", hipGetErrorString(e)); }
        const int K = 1000;
        hipEventRecord(e0);
        for (int i = 0; i < K; i++, n++) { hipLaunchKernelGGL(kernel, dim3(2048), dim3(192), 0, 0, in[n % NB], out[n % NB]); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %7.2f us per launch  %5.2f TB/s\n", name, ms * 1e3 / K, 100.663296 / (ms * 1e3 / K) );
    };
    time("R0 128x64 tiles, W0 half-block stores", tile_copy<0, 0, 0>);
    time("R0 128x64 tiles, W0 half-block stores 1500 instr apart", tile_copy<0, 0, 1500>);
    time("R0 128x64 tiles, W1 whole-block stores", tile_copy<0, 1, 0>);
    time("R1 512x16 tiles, W0 half-block stores", tile_copy<1, 0, 0>);
    time("R1 512x16 tiles, W0 half-block stores 1500 instr apart", tile_copy<1, 0, 1500>);
    time("R1 512x16 tiles, W1 whole-block stores", tile_copy<1, 1, 0>);
    time("R2 512x16 tiles with 16-byte loads, W1 whole-block stores", tile_copy<2, 1, 0>);
    time("R1 512x16 tiles, W1 whole-block stores (again)", tile_copy<1, 1, 0>);
    time("R0 128x64 tiles, W0 half-block stores (again)", tile_copy<0, 0, 0>);
    return 0;
}
