// jpeg_pixels_code.hpp — host-callable launcher of the fused pixel -> finished scan kernel (jpeg_pixels_code.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace pixo_dev {

// The launch's geometry: 512-pixel-wide tiles = groups of the scan, `images` equally sized images, every image one segment — or,
// restart intervals of whole MCU rows, one segment per `seg_rows` tile rows.
struct PixelsCodePlan {
    uint32_t units_x = 0, units_y = 0;   // MCUs (4:2:0) / blocks (4:4:4) per row, rows
    uint32_t tiles_x = 0, tiles_y = 0;
    uint32_t images = 1;
    uint32_t seg_rows = 0, segs_per_img = 1, seg_blocks64 = 0;
    uint32_t sup_copies = 1, sup_stride = 0; // the look-backs' block sums: copies (a power of two), u64 words from one copy to the next
    uint64_t groups = 0, segments = 0;   // of the whole launch
    size_t state_words = 0;              // u64 words of d_state (zero before the launch)
};
// restart_mcus: JpegOptions::restart_interval if the scan emits RSTn markers (0: none) — must be a multiple of the MCUs per row
PixelsCodePlan pixels_code_plan(uint32_t W, uint32_t H, bool s420, uint32_t images, uint32_t restart_mcus, bool gray = false); // gray: tiles of 1536 x 8 pixels (192 blocks of one block row)
// Does the kernel serve such a scan?  RGB (4:2:0 or 4:4:4), W >= 4, standard or any GIVEN tables; restart intervals only as whole
// MCU rows of one image.  (Gray images, optimised tables, bands of a multi-GPU image: coefficient kernel + scan_code + stuff_fused.)
bool pixels_code_supported(uint32_t W, uint32_t H, bool gray, bool s420, uint32_t images, uint32_t restart_mcus);

// `p.images` x `p.segs_per_img` baseline scans with the tables at d_tables (packed form followed by the flat walk's form, as
// pixo_dev::ScanArgs::tables): pixels -> colour -> DCT -> quantise -> encode_block -> BitWriterMsb's bytes (0x00 stuffed behind every
// 0xFF, every scan's last byte 1-padded) at d_out, `gap` bytes left free between two scans (rst_markers: gap = 2, and they receive
// FF D0+(n & 7)), in ONE kernel: no coefficient tuple, no packed bit stream in HBM.  Replaces launch_jpeg_coeffs + launch_scan_code +
// launch_stuff_fused and leaves their totals: host_totals[0] / d_state[1] = the last scan's length in bits (unpadded),
// host_totals[1] = the bytes from the first scan's first to the last scan's last (stuffed, gaps included), host_totals[2] = the
// last scan's bytes before stuffing, host_totals[3] / d_state[0] = the abort flag of the bounded waits; host_segs[s] (pinned, or
// null) = where scan s ends.  Nothing is stored beyond d_out[out_cap) — d_out may be device memory or host memory the GPU can
// write; the caller compares host_totals[1] with out_cap and repeats with more room.  d_state: p.state_words u64, ZERO (the
// launcher clears it unless state_is_zero); d_clear / clear_words: words this launch zeroes on the side — the state of the launch
// BEFORE it, which alternates with this one's (a context keeps two, so that no memset launch is needed between files).
// seed_dc: DC predictors of the launch's first Y / Cb / Cr block (null: zeros); pad_last = false: the last scan's last byte is
// not 1-padded (its incomplete byte is not stored: host_totals[0] says how many bits there are).
// d_block_spill: p.groups x 192 x 128 bytes of device memory (never read by the caller): groups whose bits do not fit one
// 6 KiB round (noise at q >= 90) park their quantised blocks there between the rounds' walks.
hipError_t launch_pixels_code(const void *d_px, uint32_t W, uint32_t H, bool gray, bool s420, const PixelsCodePlan &p, uint32_t gap, bool rst_markers,
                              const float *d_qt, const uint32_t *d_tables, unsigned long long *d_state, bool state_is_zero,
                              unsigned long long *d_clear, size_t clear_words, uint8_t *d_out, uint64_t out_cap, unsigned long long *host_totals,
                              unsigned long long *host_segs, const int16_t seed_dc[3], bool pad_last, void *d_block_spill, hipStream_t s,
                              uint32_t spin_budget = 1u << 20);

// Symbol statistics for optimised tables straight from the pixels (the reference's optimised-Huffman encode runs its pixel pipeline
// twice as well: count_block over every MCU, src/jpeg/mod.rs:826-860, then encode_scan): phases A and B of the fused kernel + the flat
// walk as a counter, ONE image (restart intervals of whole MCU rows as planned by pixels_code_plan), no tuple.  d_hist: kTableWords
// u64 in launch_scan_count's layout ([class][12 DC + 256 AC]); d_scratch: pixels_count_scratch_bytes(p) bytes.
size_t pixels_count_scratch_bytes(const PixelsCodePlan &p);
hipError_t launch_pixels_count(const void *d_px, uint32_t W, uint32_t H, bool gray, bool s420, const PixelsCodePlan &p, const float *d_qt, void *d_scratch,
                               unsigned long long *d_hist, hipStream_t s);

} // namespace pixo_dev
