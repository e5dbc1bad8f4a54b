// jpeg_entropy.hip — gfx950 kernels of the device entropy stage: the baseline
// Huffman scan of a coefficient tuple that is already in HBM, byte-identical to the reference's
// encode_scan + encode_block + BitWriterMsb (src/jpeg/mod.rs:1408-1563, src/jpeg/huffman.rs:
// 423-481, src/bits.rs:195-293), restart markers included.
//
//   1. lengths   one lane per block (scan order): bit length of the block            u32[n]
//   2. scan      exclusive prefix sum of the lengths -> absolute bit offset          u64[n], total
//   3. pack      one lane per block: codes + value bits at the block's offset into a zeroed
//                MSB-first word stream; the last block also writes the 1-padding of the final byte
//   4. ff count  0xFF bytes per 4 KiB tile of the packed stream, scanned like (2)
//   5. stuff     copies the stream to its final place, inserting 0x00 after every 0xFF
//   (0. count    optional: DC-category / AC run-size histograms for optimised tables)
//   (restart intervals: segment byte sizes from the prefix sum after 2, each segment packed at its
//    own byte offset in 3, FF D0+(k & 7) written over two reserved zero bytes after 5)
//
// A block is 128 contiguous bytes and a lane walks it serially (the zero-run state machine is
// inherently sequential); 64 lanes = 64 consecutive blocks.  All of it is integer work bounded
// by HBM traffic and scattered 4-byte stores: no MFMA, no LDS tiling beyond the tables.
#include <hip/hip_runtime.h>

#include "jpeg_entropy.hpp"
#include "jpeg_scan_block.h"

namespace pixo_dev {
using namespace pixo_scan;

namespace {
constexpr int kScanThreads = 256;

enum { WHAT_LENGTH = 0, WHAT_PACK = 1 }; // (symbol statistics: scan_count_kernel in jpeg_scan_fused.hip)

// first block (scan order) of restart segment k
__device__ __forceinline__ uint64_t segment_first(const ScanArgs &a, uint64_t k) { return k * a.restart * a.blocks_per_mcu; }

template <int WHAT>
__global__ __launch_bounds__(kScanThreads) void scan_blocks_kernel(const ScanArgs a, uint32_t *len, const uint64_t *off,
                                                                  uint32_t *stream, uint64_t total_bits, const uint64_t *seg_byte_off)
{
    __shared__ uint32_t tab[kTableWords];
    for (int i = threadIdx.x; i < kTableWords; i += kScanThreads) tab[i] = a.tables[i];
    __syncthreads();
    const uint64_t s = (uint64_t)blockIdx.x * kScanThreads + threadIdx.x;
    if (s < a.nblocks) {
        const BlockRef ref = block_of(a.mode, s);
        const int16_t *base = ref.comp == 0 ? a.y : (ref.comp == 1 ? a.cb : a.cr);
        const uint4 *p = reinterpret_cast<const uint4 *>(base + ref.index * 64);
        uint32_t w[32];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint4 q = p[i];
            w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
        }
        // DC predictor: the previous block of the same component; 0 for the first block of each
        // component in a restart segment (jpeg/mod.rs:1441-1444)
        int prev_dc = ref.index ? (int)base[(ref.index - 1) * 64] : (int)a.seed_dc[ref.comp];
        uint64_t mcu = 0, seg = 0;
        if (a.restart) {
            mcu = s / a.blocks_per_mcu;
            seg = mcu / a.restart;
            const uint32_t k = (uint32_t)(s - mcu * a.blocks_per_mcu);
            const bool first_of_comp = a.mode == 2 ? (k == 0 || k >= 4) : true;
            if (mcu % a.restart == 0 && first_of_comp) prev_dc = 0;
        }
        const int cls = ref.comp == 0 ? 0 : 1;
        if (WHAT == WHAT_LENGTH) {
            LengthVisitor v{tab + cls * kClassSyms, 0};
            walk_block(w, prev_dc, v);
            len[s] = v.bits;
        } else {
            PackVisitor v;
            v.tab = tab + cls * kClassSyms;
            uint64_t pos = off[s] + a.bit_base, end_bits = total_bits + a.bit_base; // end of the stream / of this block's segment
            bool last_of_segment = s == a.nblocks - 1;
            if (a.restart) { // the segment starts at its own byte offset; inside it the bits are contiguous
                const uint64_t first = segment_first(a, seg), next = segment_first(a, seg + 1);
                const uint64_t seg_bits0 = off[first];
                const uint64_t seg_end = next < a.nblocks ? off[next] : total_bits;
                pos = seg_byte_off[seg] * 8 + (pos - seg_bits0);
                end_bits = seg_byte_off[seg] * 8 + (seg_end - seg_bits0);
                last_of_segment = s + 1 == next || s == a.nblocks - 1;
            }
            v.begin(stream, pos);
            walk_block(w, prev_dc, v);
            v.finish();
            if (last_of_segment && a.pad_last) { // BitWriterMsb::flush: pad the last byte with 1-bits (bits.rs:261-272)
                const int n = (int)((8 - (end_bits & 7)) & 7);
                if (n) v.or_word(end_bits >> 5, ((1u << n) - 1u) << (32 - (int)(end_bits & 31) - n));
            }
        }
    }
}

// ---- exclusive scan of u32 -> u64 (three small kernels; tiles of 2048 elements) --------------
constexpr int kTileElems = 2048, kPerThread = kTileElems / kScanThreads;

__device__ __forceinline__ uint64_t wg_exclusive_scan(uint64_t v, uint64_t *total)
{ // 256 threads, Hillis-Steele in LDS
    __shared__ uint64_t buf[2][kScanThreads];
    int cur = 0;
    buf[0][threadIdx.x] = v;
    __syncthreads();
#pragma unroll
    for (int d = 1; d < kScanThreads; d <<= 1) {
        uint64_t x = buf[cur][threadIdx.x];
        if ((int)threadIdx.x >= d) x += buf[cur][threadIdx.x - d];
        buf[cur ^ 1][threadIdx.x] = x;
        cur ^= 1;
        __syncthreads();
    }
    const uint64_t incl = buf[cur][threadIdx.x];
    if (total) *total = buf[cur][kScanThreads - 1];
    __syncthreads();
    return incl - v;
}

__global__ __launch_bounds__(kScanThreads) void tile_sums_kernel(const uint32_t *in, uint64_t n, uint64_t *tile_sum)
{
    const uint64_t i0 = (uint64_t)blockIdx.x * kTileElems + (uint64_t)threadIdx.x * kPerThread;
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < kPerThread; k++)
        if (i0 + k < n) s += in[i0 + k];
    uint64_t total;
    (void)wg_exclusive_scan(s, &total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanThreads) void scan_tiles_kernel(uint64_t *tile_sum, uint64_t ntiles, uint64_t *grand_total)
{ // one workgroup: tile sums -> exclusive tile bases, in place
    uint64_t carry = 0;
    for (uint64_t t0 = 0; t0 < ntiles; t0 += kScanThreads) {
        const uint64_t t = t0 + threadIdx.x;
        const uint64_t v = t < ntiles ? tile_sum[t] : 0;
        uint64_t total;
        const uint64_t ex = wg_exclusive_scan(v, &total);
        if (t < ntiles) tile_sum[t] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *grand_total = carry;
}

__global__ __launch_bounds__(kScanThreads) void downsweep_kernel(const uint32_t *in, uint64_t n, const uint64_t *tile_base,
                                                                uint64_t *out)
{
    const uint64_t i0 = (uint64_t)blockIdx.x * kTileElems + (uint64_t)threadIdx.x * kPerThread;
    uint32_t v[kPerThread];
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < kPerThread; k++) { v[k] = i0 + k < n ? in[i0 + k] : 0; s += v[k]; }
    uint64_t run = tile_base[blockIdx.x] + wg_exclusive_scan(s, nullptr);
#pragma unroll
    for (int k = 0; k < kPerThread; k++) {
        if (i0 + k < n) out[i0 + k] = run;
        run += v[k];
    }
}

// ---- byte stuffing (bits.rs:245-253): tiles of 1024 words = 4096 stream bytes ------------------
constexpr int kStuffWordsPerThread = 4, kStuffTileBytes = kScanThreads * kStuffWordsPerThread * 4;

__device__ __forceinline__ uint32_t ff_bytes(uint32_t word, uint64_t first_byte, uint64_t nbytes)
{ // how many of the (at most 4) stream bytes of this big-endian word are 0xFF and exist
    uint32_t c = 0;
#pragma unroll
    for (int b = 0; b < 4; b++)
        if (first_byte + b < nbytes && ((word >> (24 - 8 * b)) & 0xFF) == 0xFF) c++;
    return c;
}

__global__ __launch_bounds__(kScanThreads) void ff_tile_count_kernel(const uint32_t *stream, uint64_t nbytes, uint32_t *tile_ff)
{
    const uint64_t w0 = ((uint64_t)blockIdx.x * kScanThreads + threadIdx.x) * kStuffWordsPerThread;
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < kStuffWordsPerThread; k++)
        if ((w0 + k) * 4 < nbytes) c += ff_bytes(stream[w0 + k], (w0 + k) * 4, nbytes);
    uint64_t total;
    (void)wg_exclusive_scan(c, &total);
    if (threadIdx.x == 0) tile_ff[blockIdx.x] = (uint32_t)total;
}

__global__ __launch_bounds__(kScanThreads) void stuff_kernel(const uint32_t *stream, uint64_t nbytes,
                                                            const uint64_t *tile_ff_base, uint8_t *out)
{
    const uint64_t w0 = ((uint64_t)blockIdx.x * kScanThreads + threadIdx.x) * kStuffWordsPerThread;
    uint32_t w[kStuffWordsPerThread];
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < kStuffWordsPerThread; k++) {
        w[k] = (w0 + k) * 4 < nbytes ? stream[w0 + k] : 0;
        c += ff_bytes(w[k], (w0 + k) * 4, nbytes);
    }
    const uint64_t before = tile_ff_base[blockIdx.x] + wg_exclusive_scan(c, nullptr);
    uint8_t *o = out + w0 * 4 + before;
#pragma unroll
    for (int k = 0; k < kStuffWordsPerThread; k++) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            if ((w0 + k) * 4 + b < nbytes) {
                const uint8_t byte = (uint8_t)(w[k] >> (24 - 8 * b));
                *o++ = byte;
                if (byte == 0xFF) *o++ = 0x00;
            }
        }
    }
}

__global__ __launch_bounds__(kScanThreads) void segment_sizes_kernel(const ScanArgs a, const uint64_t *off, const uint64_t *total_bits,
                                                                    uint64_t nsegments, uint32_t *seg_bytes)
{
    const uint64_t k = (uint64_t)blockIdx.x * kScanThreads + threadIdx.x;
    if (k >= nsegments) return;
    const uint64_t first = segment_first(a, k), next = segment_first(a, k + 1);
    const uint64_t bits = (next < a.nblocks ? off[next] : *total_bits) - off[first];
    seg_bytes[k] = (uint32_t)((bits + 7) / 8 + (k + 1 < nsegments ? a.marker_bytes : 0));
}

// 0xFF bytes of the packed stream before byte `pos` (wave-uniform), computed by a whole wavefront: the
// tile's base count + the tile's words before `pos` spread over the 64 lanes (a single thread walking up
// to 1024 words took 115 us).  Bytes sit MSB-first in the words.  Every lane returns the total.
__device__ __forceinline__ uint64_t ff_before_wave(const uint32_t *stream, const uint64_t *tile_ff_base, uint64_t pos)
{
    const uint64_t tile = pos / kStuffTileBytes, w0 = tile * (kStuffTileBytes / 4);
    const int lane = threadIdx.x & 63;
    uint32_t n = 0;
    for (uint64_t w = w0 + lane; w * 4 < pos; w += 64) {
        const uint64_t left = pos - w * 4; // bytes of this word that lie before pos (>= 1)
        uint32_t x = ~stream[w];           // a 0xFF byte becomes 0x00
        if (left < 4) x |= 0xFFFFFFFFu >> (8 * (uint32_t)left); // bytes at or after pos never count
        // exact zero-byte detector: 0x80 in every byte of x that is zero
        const uint32_t y = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
        n += (uint32_t)__builtin_popcount(y);
    }
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    return tile_ff_base[tile] + n;
}

// one wavefront per segment
__global__ __launch_bounds__(64) void segment_out_offsets_kernel(const uint64_t *seg_byte_off, uint64_t nsegments, uint64_t nbytes,
                                                                 const uint32_t *stream, const uint64_t *tile_ff_base, uint64_t *seg_out)
{
    const uint64_t k = blockIdx.x;
    const uint64_t pos = seg_byte_off[k];
    // an empty segment at the very end starts where the stream ends: the caller substitutes its size
    const uint64_t r = pos < nbytes ? pos + ff_before_wave(stream, tile_ff_base, pos) : ~0ull;
    if (threadIdx.x == 0) seg_out[k] = r;
}

__global__ __launch_bounds__(64) void restart_markers_kernel(const ScanArgs a, const uint64_t *off, const uint64_t *seg_byte_off,
                                                             uint64_t nsegments, const uint32_t *stream, const uint64_t *tile_ff_base,
                                                             uint8_t *out)
{
    const uint64_t k = blockIdx.x; // < nsegments - 1
    // the marker's two (still zero) bytes sit right before the next segment
    const uint64_t pos = seg_byte_off[k + 1] - 2;
    const uint64_t ff = ff_before_wave(stream, tile_ff_base, pos);
    if (threadIdx.x == 0) {
        out[pos + ff] = 0xFF;
        out[pos + ff + 1] = (uint8_t)(0xD0 + (k & 7)); // jpeg/mod.rs:1436-1439
    }
}

// ---- progressive scans ---------------------------------------------------------------------------
struct ProgBlock { int scan; uint32_t w[32]; int prev_dc; };
__device__ __forceinline__ ProgLayout layout_of(const ProgArgs &a)
{
    ProgLayout l;
#pragma unroll
    for (int i = 0; i < 8; i++) l.first[i] = a.first[i];
    return l;
}
__device__ __forceinline__ void load_prog_block(const ProgArgs &a, const ProgLayout &l, uint64_t v, ProgBlock &b)
{
    b.scan = prog_scan_of(l, v);
    const int comp = prog_comp(b.scan);
    const int16_t *base = comp == 0 ? a.y : (comp == 1 ? a.cb : a.cr);
    const uint64_t i = v - l.first[b.scan];
    const uint4 *p = reinterpret_cast<const uint4 *>(base + i * 64);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint4 q = p[k];
        b.w[4 * k] = q.x; b.w[4 * k + 1] = q.y; b.w[4 * k + 2] = q.z; b.w[4 * k + 3] = q.w;
    }
    b.prev_dc = i ? (int)base[(i - 1) * 64] : 0;
}

__global__ __launch_bounds__(kScanThreads) void prog_flags_kernel(const ProgArgs a)
{
    const uint64_t v = (uint64_t)blockIdx.x * kScanThreads + threadIdx.x;
    if (v >= a.first[7]) return;
    const ProgLayout l = layout_of(a);
    ProgBlock b;
    load_prog_block(a, l, v, b);
    uint32_t f = 0;
    if (b.scan >= 3) {
        const int band = prog_band(b.scan);
        f = band == 0 ? band_flags<1, 10>(b.w) : (band == 1 ? band_flags<11, 63>(b.w) : band_flags<1, 63>(b.w));
    }
    a.flags[v] = f;
    a.nonempty[v] = f & 1u;
}

__global__ __launch_bounds__(kScanThreads) void prog_by_rank_kernel(const ProgArgs a)
{
    const uint64_t v = (uint64_t)blockIdx.x * kScanThreads + threadIdx.x;
    if (v >= a.first[7]) return;
    if (a.flags[v] & 1u) a.by_rank[a.rank[v]] = (uint32_t)v;
}

template <int WHAT>
__global__ __launch_bounds__(kScanThreads) void prog_blocks_kernel(const ProgArgs a, uint32_t *len, const uint64_t *off, uint64_t total_bits,
                                                                  const uint64_t *seg_byte_off, uint32_t *stream)
{
    __shared__ uint32_t tab[kTableWords];
    for (int i = threadIdx.x; i < kTableWords; i += kScanThreads) tab[i] = a.tables[i];
    __syncthreads();
    const uint64_t v = (uint64_t)blockIdx.x * kScanThreads + threadIdx.x;
    if (v >= a.first[7]) return;
    const ProgLayout l = layout_of(a);
    ProgBlock b;
    load_prog_block(a, l, v, b);
    const uint64_t first = l.first[b.scan], next = l.first[b.scan + 1];
    const uint32_t flags = a.flags[v];
    const uint32_t before = b.scan >= 3 ? band_run_before(v, first, a.rank[v], a.rank[first], a.by_rank, a.flags) : 0u;
    const bool last = v + 1 == next;
    const uint32_t *t = tab + (prog_comp(b.scan) ? 1 : 0) * kClassSyms;
    if (WHAT == WHAT_LENGTH) {
        LengthVisitor vis{t, 0};
        prog_emit(b.scan, b.w, b.prev_dc, flags, before, last, vis);
        len[v] = vis.bits;
    } else {
        PackVisitor vis;
        vis.tab = t;
        const uint64_t base_bits = seg_byte_off[b.scan] * 8, scan_bits0 = off[first];
        vis.begin(stream, base_bits + (off[v] - scan_bits0));
        prog_emit(b.scan, b.w, b.prev_dc, flags, before, last, vis);
        vis.finish();
        if (last) { // every scan has its own BitWriterMsb: flush pads with 1-bits (jpeg/mod.rs:925)
            const uint64_t end_bits = base_bits + ((next < a.first[7] ? off[next] : total_bits) - scan_bits0);
            const int n = (int)((8 - (end_bits & 7)) & 7);
            if (n) vis.or_word(end_bits >> 5, ((1u << n) - 1u) << (32 - (int)(end_bits & 31) - n));
        }
    }
}

__global__ void prog_segment_sizes_kernel(const ProgArgs a, const uint64_t *off, const uint64_t *total_bits, uint32_t *seg_bytes)
{
    const int i = threadIdx.x;
    if (i >= 7) return;
    const uint64_t first = a.first[i], next = a.first[i + 1];
    const uint64_t end = next < a.first[7] ? off[next] : *total_bits;
    const uint64_t begin = first < a.first[7] ? off[first] : *total_bits;
    seg_bytes[i] = (uint32_t)((end - begin + 7) / 8);
}

inline unsigned grid_for(uint64_t n, uint64_t per_group) { return (unsigned)((n + per_group - 1) / per_group); }
} // namespace

size_t scan_tile_count(uint64_t n) { return (size_t)((n + kTileElems - 1) / kTileElems); }
size_t stuff_tile_count(uint64_t nbytes) { return (size_t)((nbytes + kStuffTileBytes - 1) / kStuffTileBytes); }

hipError_t launch_scan_lengths(const ScanArgs &a, uint32_t *d_len, hipStream_t s)
{
    hipLaunchKernelGGL((scan_blocks_kernel<WHAT_LENGTH>), dim3(grid_for(a.nblocks, kScanThreads)), dim3(kScanThreads), 0, s, a,
                       d_len, nullptr, nullptr, 0, nullptr);
    return hipGetLastError();
}

hipError_t launch_scan_pack(const ScanArgs &a, const uint64_t *d_off, uint64_t total_bits, const SegmentPlan *seg,
                            uint32_t *d_stream, hipStream_t s)
{
    hipLaunchKernelGGL((scan_blocks_kernel<WHAT_PACK>), dim3(grid_for(a.nblocks, kScanThreads)), dim3(kScanThreads), 0, s, a,
                       nullptr, d_off, d_stream, total_bits, seg ? seg->seg_byte_off : nullptr);
    return hipGetLastError();
}

hipError_t launch_segment_sizes(const ScanArgs &a, const uint64_t *d_off, const uint64_t *d_total_bits, uint64_t nsegments,
                                uint32_t *d_seg_bytes, hipStream_t s)
{
    hipLaunchKernelGGL(segment_sizes_kernel, dim3(grid_for(nsegments, kScanThreads)), dim3(kScanThreads), 0, s, a, d_off, d_total_bits,
                       nsegments, d_seg_bytes);
    return hipGetLastError();
}

hipError_t launch_prog_flags(const ProgArgs &a, hipStream_t s)
{
    if (a.first[7] == 0) return hipSuccess;
    hipLaunchKernelGGL(prog_flags_kernel, dim3(grid_for(a.first[7], kScanThreads)), dim3(kScanThreads), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_prog_by_rank(const ProgArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(prog_by_rank_kernel, dim3(grid_for(a.first[7], kScanThreads)), dim3(kScanThreads), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_prog_lengths(const ProgArgs &a, uint32_t *d_len, hipStream_t s)
{
    hipLaunchKernelGGL((prog_blocks_kernel<WHAT_LENGTH>), dim3(grid_for(a.first[7], kScanThreads)), dim3(kScanThreads), 0, s, a, d_len,
                       nullptr, 0, nullptr, nullptr);
    return hipGetLastError();
}
hipError_t launch_prog_segment_sizes(const ProgArgs &a, const uint64_t *d_off, const uint64_t *d_total_bits, uint32_t *d_seg_bytes,
                                     hipStream_t s)
{
    hipLaunchKernelGGL(prog_segment_sizes_kernel, dim3(1), dim3(64), 0, s, a, d_off, d_total_bits, d_seg_bytes);
    return hipGetLastError();
}
hipError_t launch_prog_pack(const ProgArgs &a, const uint64_t *d_off, uint64_t total_bits, const uint64_t *d_seg_byte_off,
                            uint32_t *d_stream, hipStream_t s)
{
    hipLaunchKernelGGL((prog_blocks_kernel<WHAT_PACK>), dim3(grid_for(a.first[7], kScanThreads)), dim3(kScanThreads), 0, s, a, nullptr,
                       d_off, total_bits, d_seg_byte_off, d_stream);
    return hipGetLastError();
}

hipError_t launch_segment_out_offsets(const SegmentPlan &seg, uint64_t nbytes, const uint32_t *d_stream, const uint64_t *d_tile_ff_base,
                                      uint64_t *d_seg_out, hipStream_t s)
{
    hipLaunchKernelGGL(segment_out_offsets_kernel, dim3((unsigned)seg.nsegments), dim3(64), 0, s,
                       seg.seg_byte_off, seg.nsegments, nbytes, d_stream, d_tile_ff_base, d_seg_out);
    return hipGetLastError();
}

hipError_t launch_restart_markers(const ScanArgs &a, const uint64_t *d_off, const SegmentPlan &seg, const uint32_t *d_stream,
                                  const uint64_t *d_tile_ff_base, uint8_t *d_out, hipStream_t s)
{
    if (seg.nsegments < 2 || a.marker_bytes == 0) return hipSuccess;
    hipLaunchKernelGGL(restart_markers_kernel, dim3((unsigned)(seg.nsegments - 1)), dim3(64), 0, s, a, d_off,
                       seg.seg_byte_off, seg.nsegments, d_stream, d_tile_ff_base, d_out);
    return hipGetLastError();
}

hipError_t launch_exclusive_scan(const uint32_t *d_in, uint64_t n, uint64_t *d_out, uint64_t *d_tile_tmp, uint64_t *d_total,
                                 hipStream_t s)
{
    const unsigned tiles = (unsigned)scan_tile_count(n);
    if (tiles == 0) return hipMemsetAsync(d_total, 0, sizeof(uint64_t), s); // nothing to sum (an empty band)
    hipLaunchKernelGGL(tile_sums_kernel, dim3(tiles), dim3(kScanThreads), 0, s, d_in, n, d_tile_tmp);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(kScanThreads), 0, s, d_tile_tmp, (uint64_t)tiles, d_total);
    if (d_out) hipLaunchKernelGGL(downsweep_kernel, dim3(tiles), dim3(kScanThreads), 0, s, d_in, n, d_tile_tmp, d_out);
    return hipGetLastError();
}

hipError_t launch_ff_tile_count(const uint32_t *d_stream, uint64_t nbytes, uint32_t *d_tile_ff, hipStream_t s)
{
    hipLaunchKernelGGL(ff_tile_count_kernel, dim3((unsigned)stuff_tile_count(nbytes)), dim3(kScanThreads), 0, s, d_stream, nbytes,
                       d_tile_ff);
    return hipGetLastError();
}

hipError_t launch_stuff(const uint32_t *d_stream, uint64_t nbytes, const uint64_t *d_tile_ff_base, uint8_t *d_out, hipStream_t s)
{
    hipLaunchKernelGGL(stuff_kernel, dim3((unsigned)stuff_tile_count(nbytes)), dim3(kScanThreads), 0, s, d_stream, nbytes,
                       d_tile_ff_base, d_out);
    return hipGetLastError();
}

} // namespace pixo_dev
