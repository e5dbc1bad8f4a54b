// NOT BUILT.  Round-3 experiment, kept for the record (profiles/r03_png_pipelined_kernel_experiment.txt):
// the PNG filter kernel as a persistent workgroup with three rows in registers.  It was written on the theory that the
// one-row kernel's load / score / store phases were correlated across the chip; the cycle counters then showed the VALU
// about 70 % busy in either form, and this form — 124 registers, 4 waves per SIMD, two barriers and ~48 register moves per
// row — is SLOWER: 60.7 us at 4 rows per workgroup, 56.2 at 2, 70.1 at 8, against 49.0 for one row per workgroup.
// The fragment below sat in pixo_amd/csrc/png_filter.hip between stage_row_regs / flush_stage_body and launch_bpp.
// The pipelined form of the register-resident kernel (round 3; rows of at most ITERS x 4 KiB, adaptive strategies,
// row base and row length multiples of 4).
// One workgroup per row had every resident workgroup of the chip loading at the same time, then scoring at the same
// time, then storing: with two generations of workgroups per 4096-row image the three phases hardly overlapped
// (loads ~7 us + VALU ~15 us + stores ~6 us per generation against 31 us of VALU work in total).  Here a workgroup
// walks down a.rows_per_wg consecutive rows with three rows in registers — A the row above, B the row being filtered,
// C the row below, whose loads are issued BEFORE B is scored and are first touched a whole row of arithmetic later — so
// the memory pipe and the VALU of one workgroup work at the same time, and every row is loaded once instead of twice.
// Per row: loads of C | scores -> wave butterflies -> LDS atomics | barrier | decision, winning filter -> LDS stage,
// checksum terms | barrier | A <- B <- C (the only wait on memory: loads a row old; the previous row's stores) |
// aligned 16-byte stores of the staged row, left in flight.  Nothing is computed on a loaded value before that point
// (addresses are clamped instead of values selected; the zeros left of the row's first byte are put in at use), and
// the accumulators alternate between two sets (row parity): the set of row r is zeroed again after its last reader
// and is next written after the following row's first barrier.
template <int BPP> __device__ __forceinline__ void load_six_clamped(const uint8_t *row, int k0, int ndw, uint32_t *x)
{
    if (k0 + 4 <= ndw) {
        if (BPP <= 4) x[1] = *reinterpret_cast<const uint32_t *>(row + 4 * (k0 >= 1 ? k0 - 1 : 0)); // (x[0] is never looked at)
        else {
            const uint2 l = *reinterpret_cast<const uint2 *>(row + 4 * (k0 >= 2 ? k0 - 2 : 0));
            x[0] = l.x; x[1] = l.y;
        }
        const uint4 c = *reinterpret_cast<const uint4 *>(row + 4 * k0);
        x[2] = c.x; x[3] = c.y; x[4] = c.z; x[5] = c.w;
    } else if (k0 < ndw) { // the row's last, partial group: dwords past the end re-read the last one (masked at use)
#pragma unroll
        for (int i = 0; i < 6; i++) {
            int k = k0 - 2 + i;
            k = k < 0 ? 0 : (k < ndw ? k : ndw - 1);
            asm volatile("" : "+v"(k)); // (keeps this path apart: merged with the other, BOTH become six dword loads)
            x[i] = *reinterpret_cast<const uint32_t *>(row + 4 * k);
        }
    }
}

// (registers: 124 / 92 / 114 for 4 / 2 / 1 groups per thread — 4 workgroups per CU; asking the allocator for 5 or 6
// waves per SIMD spills)
template <int BPP, int ITERS>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4))) void png_filter_pipe_kernel(const Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
    __shared__ unsigned long long acc64[2][2];
    __shared__ unsigned int acc32[2][8];
    // chunks of rows_per_wg rows; 8 consecutive chunks on one XCD (the row above a chunk's first row is the previous
    // chunk's last: same L2), the groups of 8 round-robin over the XCDs (png_filter_kernel explains why not bands)
    const uint32_t nb = gridDim.x, full = nb & ~63u;
    uint32_t chunk = blockIdx.x;
    if (chunk < full) { const uint32_t xcd = chunk & 7u, i = chunk >> 3; chunk = (((i >> 3) * 8u + xcd) << 3) + (i & 7u); }
    const uint32_t y0 = chunk * a.rows_per_wg;
    const uint32_t rows = a.height - y0 < a.rows_per_wg ? a.height - y0 : a.rows_per_wg;
    const int n = (int)a.row_bytes;
    const int ndw = n / 4, per_iter = kThreads * 4;
    const int strategy = a.strategy;
    const bool fast = strategy == PNG_S_ADAPTIVE_FAST;
    const uint8_t *row = a.data + (size_t)y0 * a.row_bytes;

    if (threadIdx.x < 16) acc32[threadIdx.x >> 3][threadIdx.x & 7] = 0;
    if (threadIdx.x < 4) acc64[threadIdx.x >> 1][threadIdx.x & 1] = 0;
    uint32_t A[ITERS][6], B[ITERS][6], C[ITERS][6];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int k0 = (int)threadIdx.x * 4 + it * per_iter;
#pragma unroll
        for (int i = 0; i < 6; i++) { A[it][i] = 0; B[it][i] = 0; C[it][i] = 0; }
        if (y0) load_six_clamped<BPP>(row - a.row_bytes, k0, ndw, A[it]);
        load_six_clamped<BPP>(row, k0, ndw, B[it]);
    }
    __syncthreads(); // the zeroed accumulators
    // (the first two rows have arrived HERE: a load still pending at the loop's head would make every later pass
    // through it wait for the stores of the row before)
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = BPP <= 4 ? 1 : 0; i < 6; i++) { asm volatile("" : "+v"(A[it][i])); asm volatile("" : "+v"(B[it][i])); }
    }

#pragma unroll 1
    for (uint32_t r = 0; r < rows; r++, row += a.row_bytes) {
        const uint32_t y = y0 + r, p = r & 1u;
        // (opaque: everything derived from the thread index — addresses, byte masks, checksum weights — would
        // otherwise be hoisted out of the row loop and held in ~60 registers across it)
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (r + 1 < rows) { // the row below: in flight while this row is scored, filtered and staged
#pragma unroll
            for (int it = 0; it < ITERS; it++) load_six_clamped<BPP>(row + a.row_bytes, tid * 4 + it * per_iter, ndw, C[it]);
        }
        Raw raw[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int i = 0; i < 6; i++) { raw[it].x[i] = B[it][i]; raw[it].u[i] = A[it][i]; }
        }
        if (tid == 0) { raw[0].x[0] = 0; raw[0].x[1] = 0; raw[0].u[0] = 0; raw[0].u[1] = 0; } // left of the row's first byte
        uint32_t sc[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int k0 = tid * 4 + it * per_iter;
            if (k0 >= ndw) continue;
            if (k0 + 4 <= ndw) score_group<BPP, false>(raw[it], k0, n, fast, sc);
            else score_group<BPP, true>(raw[it], k0, n, fast, sc);
        }
        // row totals (< 2^32: at most 16 KiB x 128): 32-bit butterflies, one LDS atomic per wavefront and score
#pragma unroll
        for (int i = 0; i < 5; i++) {
            if (fast && (i == F_NONE || i == F_AVG)) continue;
            const uint32_t t = wave_sum_u32(sc[i]);
            if ((tid & 63) == 0) atomicAdd(&acc32[p][i], t);
        }
        __syncthreads();
        unsigned long long tot[5];
#pragma unroll
        for (int i = 0; i < 5; i++) tot[i] = acc32[p][i];
        const int f = decide(strategy, tot, (unsigned long long)n);
        if (a.winner0 && y == 0 && tid == 0) *a.winner0 = f;
        switch (f) {
        case F_NONE: stage_row_regs<BPP, F_NONE, kThreads, ITERS>(a, y, raw, n, acc64[p], tid); break;
        case F_SUB: stage_row_regs<BPP, F_SUB, kThreads, ITERS>(a, y, raw, n, acc64[p], tid); break;
        case F_UP: stage_row_regs<BPP, F_UP, kThreads, ITERS>(a, y, raw, n, acc64[p], tid); break;
        case F_AVG: stage_row_regs<BPP, F_AVG, kThreads, ITERS>(a, y, raw, n, acc64[p], tid); break;
        default: stage_row_regs<BPP, F_PAETH, kThreads, ITERS>(a, y, raw, n, acc64[p], tid); break;
        }
        __syncthreads(); // (every reader of this parity's score totals is done, the checksum totals are complete)
        // A <- B <- C before this row's stores are issued: the wait that belongs to these moves then covers only
        // loads issued a row ago and the PREVIOUS row's stores
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int i = BPP <= 4 ? 1 : 0; i < 6; i++) { // (dword 0 is only looked at with 6 or 8 bytes per pixel)
                A[it][i] = B[it][i]; B[it][i] = C[it][i];
                asm volatile("" : "+v"(B[it][i]));
            }
        }
        flush_stage_body<kThreads>(stage, a.out + (size_t)y * (a.row_bytes + 1), n, tid);
        if (tid == 0) {
            a.row_sums[2 * (size_t)y] = acc64[p][0]; a.row_sums[2 * (size_t)y + 1] = acc64[p][1];
            acc64[p][0] = 0; acc64[p][1] = 0;
        }
        if (tid < 8) acc32[p][tid] = 0;
    }
}

template <int BPP, int ITERS> void launch_pipe(const Args &a, uint32_t chunks, hipStream_t s)
{
    hipLaunchKernelGGL((png_filter_pipe_kernel<BPP, ITERS>), dim3(chunks), dim3(kThreads), a.stage_bytes, s, a);
}

