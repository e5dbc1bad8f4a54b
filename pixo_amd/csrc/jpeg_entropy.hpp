// jpeg_entropy.hpp — host-callable launchers of the device entropy stage (jpeg_entropy.hip).
// All pointers are device pointers; every launcher only enqueues work on `stream`.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace pixo_dev {

struct ScanArgs {
    const int16_t *y, *cb, *cr; // coefficient tuple, natural order, 64 i16 per block
    const uint32_t *tables;     // pixo_scan::kTableWords words: (length << 16) | code; behind them pixo_scan::kWalkWords words in the flat walk's form
    int mode;                   // 0 gray, 1 4:4:4, 2 4:2:0 (block order of encode_scan)
    uint64_t nblocks;           // blocks in scan order
    uint32_t restart;           // MCUs per segment, 0 = one uninterrupted stream.  A segment starts on a byte
                                // boundary with DC predictors 0: a restart interval (jpeg/mod.rs:1423-1445)
                                // or, for a batch, one whole image
    uint32_t blocks_per_mcu;    // 1, 3 or 6
    uint32_t marker_bytes;      // 2 = RSTn marker after every segment but the last; 0 = batch of images
    // A band of a larger image (one uninterrupted stream, restart == 0):
    int16_t seed_dc[3];         // DC predictors of the first Y / Cb / Cr block: the last DCs of the band above (0: a whole image)
    uint32_t bit_base;          // the stream's first bit is bit `bit_base` of d_stream (< 64; 0: a whole image)
    uint32_t pad_last;          // 1 = the last block pads the final byte with 1-bits (BitWriterMsb::flush); 0 = a band leaves its
                                //     tail bits for the splice
};

// Scans with restart markers: a segment is `restart` MCUs; every segment starts on a byte boundary
// (1-padding), DC predictors restart at 0, and all but the last are followed by FF D0+(k & 7).
struct SegmentPlan {
    uint64_t nsegments;
    const uint64_t *seg_byte_off; // [nsegments] byte offset of each segment in the packed stream
};

size_t scan_tile_count(uint64_t n);        // u64 scratch words launch_exclusive_scan needs for n elements
size_t stuff_tile_count(uint64_t nbytes);  // 4 KiB tiles of the packed stream

// symbol statistics of a scan (restart intervals honoured; jpeg_scan_fused.hip).  d_hist: pixo_scan::kTableWords 64-bit
// counters ([class][12 DC + 256 AC]), overwritten; d_scratch: scan_count_scratch_bytes() bytes.
size_t scan_count_scratch_bytes();
hipError_t launch_scan_count(const ScanArgs &a, uint32_t *d_scratch, unsigned long long *d_hist, hipStream_t s);
hipError_t launch_scan_lengths(const ScanArgs &a, uint32_t *d_len, hipStream_t s);
// d_out[i] = sum of d_in[0..i) (may be null: totals only); *d_total = sum of all
hipError_t launch_exclusive_scan(const uint32_t *d_in, uint64_t n, uint64_t *d_out, uint64_t *d_tile_tmp, uint64_t *d_total,
                                 hipStream_t s);
// d_stream: zeroed, at least total_bits / 32 + 2 words.  seg: null without restart markers.
hipError_t launch_scan_pack(const ScanArgs &a, const uint64_t *d_off, uint64_t total_bits, const SegmentPlan *seg,
                            uint32_t *d_stream, hipStream_t s);
// Restart segments: d_seg_bytes[k] = bytes of segment k in the packed stream (padding and, except for
// the last, the two marker bytes included), from the exclusive bit offsets d_off and the total.
hipError_t launch_segment_sizes(const ScanArgs &a, const uint64_t *d_off, const uint64_t *d_total_bits, uint64_t nsegments,
                                uint32_t *d_seg_bytes, hipStream_t s);
// d_seg_out[k] = offset of segment k in the STUFFED stream, k < nsegments (~0 for an empty segment that
// starts at the end of the packed stream, nbytes): where each image of a batch / each progressive scan
// begins.  After launch_ff_tile_count + its scan.
hipError_t launch_segment_out_offsets(const SegmentPlan &seg, uint64_t nbytes, const uint32_t *d_stream,
                                      const uint64_t *d_tile_ff_base, uint64_t *d_seg_out, hipStream_t s);
// After launch_stuff: writes FF D0+(k & 7) over the two zero bytes that follow segment k < nsegments - 1.
hipError_t launch_restart_markers(const ScanArgs &a, const uint64_t *d_off, const SegmentPlan &seg, const uint32_t *d_stream,
                                  const uint64_t *d_tile_ff_base, uint8_t *d_out, hipStream_t s);
hipError_t launch_ff_tile_count(const uint32_t *d_stream, uint64_t nbytes, uint32_t *d_tile_ff, hipStream_t s);
// d_out: nbytes + (number of 0xFF bytes) bytes
hipError_t launch_stuff(const uint32_t *d_stream, uint64_t nbytes, const uint64_t *d_tile_ff_base, uint8_t *d_out, hipStream_t s);

// ---- one uninterrupted baseline scan in two single-pass kernels (jpeg_scan_fused.hip) -------------------------
// code: tuple -> packed bit stream starting at bit 0 of d_stream (a.restart must be 0, a.bit_base ignored; a.seed_dc and
// a.pad_last as above).  d_state: fused_code_state_words(nblocks) u64 (zeroed by the launcher); afterwards d_state[1] =
// the scan's length in bits.  d_stream: room for nblocks * 209 + 64 bytes (a block has at most 1665 bits).
size_t fused_code_state_words(uint64_t nblocks);
// A scan coded in PIECES — runs of consecutive blocks, one launch pair (code, stuff) each, so that the first pieces' bytes
// can travel to the host while the later ones are coded.  Piece k codes blocks [first_block, first_block + a.nblocks)
// of the tuple `a` describes; chain[k] = bits of the scan before piece k (the code kernel of piece k reads chain[k] and
// chain[k - 1], k > 0, and writes chain[k + 1]); its stream starts with the chain[k] % 8 last bits of the piece before
// (taken from prev_stream), i.e. it is byte-aligned with the scan: stuffed with shift 0, `band` = true for every piece but
// the last (whole bytes only; the leftover bits lead the next piece), a.pad_last = 1 only for the last.
struct ScanPiece {
    uint64_t first_block;
    uint32_t index;            // k
    unsigned long long *chain; // device, pieces + 1 words (null: not a chain)
    const uint32_t *prev_stream;
};
// SEGMENTED scans in the single-pass kernels: the scan is cut into byte-aligned segments of `blocks` blocks (the last
// one may be shorter) — the images of a batch (marker_bytes = room for the next file's headers and this file's EOI) or
// restart intervals (marker_bytes 2, rst_markers: FF D0+(k & 7) behind every segment but the last, jpeg/mod.rs:1423-1445).  Every segment starts from DC predictors 0, is packed into a
// stream of its own (region `stream_words` * k of d_stream: blocks * 209 bytes + slack, a multiple of 16) and 1-padded.
// code<SEG> -> bits[k]; launch_seg_layout -> layout / bytes; stuff<SEG> -> the segments back to back in d_out (markers
// in between), out_end[k] = where segment k's entropy-coded bytes end (device array; host_out_end: the same in pinned
// host memory, or null).
struct SegArgs {
    uint64_t nsegs = 0;
    uint32_t blocks = 0, groups = 0;     // per segment: blocks, groups of 192 (seg_groups(blocks))
    uint64_t stream_words = 0;
    unsigned long long *bits = nullptr;        // [nsegs] out of code
    const unsigned long long *layout = nullptr; // [nsegs + 2] out of launch_seg_layout: [0] tiles, [1 + k] first tile of segment k
    const unsigned long long *bytes = nullptr;  // [nsegs] packed bytes per segment
    unsigned long long *out_end = nullptr;      // [nsegs]
    unsigned long long *host_out_end = nullptr;
    uint32_t marker_bytes = 0;  // bytes left free in d_out behind every segment but the last (at most seg_max_gap())
    uint32_t rst_markers = 0;   // 1: marker_bytes == 2 and the stuffing kernel writes FF D0+(k & 7) there
    // Segments of DIFFERENT sizes (the scans of a progressive file, prog_code_kernel): var != 0, nsegs <= 8, segment k's
    // packed stream begins at word var_word[k] of d_stream (blocks / groups / stream_words are not used then)
    uint32_t var = 0;
    uint64_t var_word[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
uint32_t seg_groups(uint64_t seg_blocks);
uint32_t seg_max_gap();
size_t fused_code_state_words_seg(uint64_t nsegs, uint64_t seg_blocks);
// headers + EOI markers of batch files that stay on the device; d_meta: [offsets[0..batch-1], end] (u64), then `hdr` header bytes
hipError_t launch_batch_seams(uint8_t *d_arena, const unsigned long long *d_meta, uint32_t batch, uint32_t hdr, hipStream_t s);
hipError_t launch_seg_layout(const SegArgs &seg, unsigned long long *d_layout, unsigned long long *d_bytes, unsigned long long *host_totals,
                             hipStream_t s); // host_totals[2] (or null) receives the total number of 16 KiB tiles

// The single-pass kernels wait for lower-numbered workgroups (look-back, shared words).  Every wait is bounded by
// `spin_budget` polls: a kernel that exhausts it raises an abort flag — d_state[0] and host_totals[3] — and ends without
// hanging the GPU; its outputs are garbage then and the caller must code the scan another way.
// state_is_zero: the caller knows d_state holds zeros (word 1 aside) — the stuffing kernel of the previous scan left it so —
// and no memset is launched.  d_clear / clear_words: words this kernel zeroes on the side (the state of the stuffing launch
// that follows), or null.  host_totals: pinned host memory (or null): [0] also receives the scan's length — no read-back copy.
hipError_t launch_scan_code(const ScanArgs &a, unsigned long long *d_state, bool state_is_zero, uint32_t *d_stream,
                            unsigned long long *d_clear, size_t clear_words, unsigned long long *host_totals, hipStream_t s,
                            const ScanPiece *piece = nullptr, const SegArgs *seg = nullptr, uint32_t spin_budget = 1u << 20);
// stuff: the stream's bytes from bit `shift` (< 8) on -> d_out with 0x00 behind every 0xFF.  band = false: all bytes of a
// whole scan (shift 0); band = true: only the whole bytes behind the band's first `shift` bits.  Reads the scan's length
// from d_code_state[1] (no host round trip) and zeroes the rest of d_code_state (code_state_words) for the next scan.
// d_state: fused_stuff_state_words(max_stream_bytes) u64, zeroed by the launcher unless state_is_zero; afterwards
// d_state[1] = bytes produced (bytes beyond out_cap are not written: the caller grows d_out and repeats this launch),
// d_state[2] = bytes of the packed stream consumed.
// One workgroup per tile: tiles [first_tile, first_tile + tiles) of stuff_tiles(stream bytes); the caller launches a
// guess, reads d_state[2] back and, if the stream has more tiles, launches the rest (first_tile > 0 keeps the state).
// host_totals (pinned host memory, or null): [1] and [2] receive d_state[1] and d_state[2] as well.
size_t fused_stuff_state_words(uint64_t max_stream_bytes);
uint64_t stuff_tiles(uint64_t stream_bytes);
uint64_t stuff_tile_bytes(); // bytes of the packed stream one stuffing workgroup takes
hipError_t launch_stuff_fused(const uint32_t *d_stream, unsigned long long *d_code_state, size_t code_state_words, uint32_t shift, bool band,
                              uint64_t max_stream_bytes, uint64_t first_tile, uint64_t tiles, unsigned long long *d_state,
                              bool state_is_zero, uint8_t *d_out, uint64_t out_cap, unsigned long long *host_totals, hipStream_t s,
                              unsigned long long *d_out_chain = nullptr, uint32_t piece = 0, const SegArgs *seg = nullptr,
                              uint32_t spin_budget = 1u << 20);
// (d_out_chain, piece: the piece's bytes go to d_out + d_out_chain[piece] (piece 0: d_out); d_out_chain[piece + 1] receives
// where they end; d_state[1] / host_totals[1] count this piece's bytes only)

// ---- progressive scans in ONE pass (round 4): the seven scans of simple_progressive_script (progressive.rs:98-110) as
// byte-aligned segments of one launch of prog_code_kernel (jpeg_scan_fused.hip) — every lane codes one (scan, block) pair
// with the flat walk, the end-of-band run counter travels through wavefront ballots and a look-back of its own, the bits
// are placed like scan_code's.  Scans without blocks (the chroma scans of a gray image) are left out: nscans <= 7.
// code -> seg.bits[k]; launch_seg_layout; stuff<SEG> with seg.var = 1 (the segments' streams at seg.var_word[k]).
struct ProgCode {
    const int16_t *y, *cb, *cr;
    const uint32_t *tables;   // packed ((length << 16) | code, absent symbols (4 << 16)), then the flat walk's form
    uint32_t nscans;
    uint32_t scan_id[7];      // index into the script: 0..2 DC of Y, Cb, Cr; 3 Y 1..10; 4 Y 11..63; 5 Cb 1..63; 6 Cr 1..63
    uint64_t size[7];         // blocks
    uint64_t first_group[8];  // scan k's first slot in the per-(scan, group) state arrays (groups of 192 blocks); [nscans] = slots in all
    // round 5: the WORKGROUPS are per component — group i of component c holds its blocks [192 i, 192 i + 192) and codes them once
    // per scan of the component (comp_pass[c][0 .. comp_npass[c]): indices k into scan_id / size / first_group, script order)
    uint64_t comp_first[4];   // first workgroup of the i-th run; [3] = workgroups in all (a component without blocks: none)
    uint32_t comp_order[3];   // the component whose groups form the i-th run
    uint64_t comp_blocks[3];
    uint32_t comp_npass[3];
    uint32_t comp_pass[3][3];
};
void prog_code_plan(ProgCode &a); // comp_* from nscans / scan_id / size (the caller has filled those and first_group)
size_t prog_code_state_words(uint64_t groups); // u64 words of d_state
uint64_t prog_groups(uint32_t scan_id, uint64_t blocks); // state slots of one scan: its component's groups of 192 blocks
// d_state: zeroed (by the launcher unless state_is_zero); d_stream: scan k at word seg.var_word[k], room for
// prog_stream_bytes(scan_id, blocks) bytes; d_clear / clear_words, host_totals ([3]: abort flag), spin_budget: as launch_scan_code
size_t prog_stream_bytes(uint32_t scan_id, uint64_t blocks);
hipError_t launch_prog_code(const ProgCode &a, const SegArgs &seg, unsigned long long *d_state, bool state_is_zero, uint32_t *d_stream,
                            unsigned long long *d_clear, size_t clear_words, unsigned long long *host_totals, hipStream_t s,
                            uint32_t spin_budget = 1u << 20);

// ---- progressive scans on the device, multi-pass (round 1; the fallback of the single-pass form) ------
struct ProgArgs {
    const int16_t *y, *cb, *cr; // coefficient tuple
    const uint32_t *tables;     // packed like ScanArgs::tables; absent symbols hold (4 << 16): code 0, 4 bits
    uint64_t first[8];          // virtual block index where scan i starts; first[7] = total
    uint32_t *flags;            // [total] per virtual block: bit 0 band not empty, bit 1 ends before the band's end
    uint32_t *nonempty;         // [total] flags & 1 (input of the rank prefix sum)
    const uint64_t *rank;       // [total] exclusive prefix sum of nonempty
    uint32_t *by_rank;          // [total] virtual index of the r-th non-empty block
};
hipError_t launch_prog_flags(const ProgArgs &a, hipStream_t s);   // fills flags, nonempty
hipError_t launch_prog_by_rank(const ProgArgs &a, hipStream_t s); // fills by_rank (after the rank prefix sum)
hipError_t launch_prog_lengths(const ProgArgs &a, uint32_t *d_len, hipStream_t s);
// d_off: exclusive bit offsets of the virtual blocks, d_total: their sum; d_seg_bytes[7]: bytes of each scan's
// packed (unstuffed, 1-padded) stream
hipError_t launch_prog_segment_sizes(const ProgArgs &a, const uint64_t *d_off, const uint64_t *d_total_bits, uint32_t *d_seg_bytes,
                                     hipStream_t s);
// d_seg_byte_off[7]: exclusive prefix sum of d_seg_bytes; d_stream zeroed
hipError_t launch_prog_pack(const ProgArgs &a, const uint64_t *d_off, uint64_t total_bits, const uint64_t *d_seg_byte_off,
                            uint32_t *d_stream, hipStream_t s);

} // namespace pixo_dev
