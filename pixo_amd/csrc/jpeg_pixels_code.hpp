// jpeg_pixels_code.hpp — host-callable launcher of the fused pixel -> packed bit stream kernel (jpeg_pixels_code.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace pixo_dev {

// One RGB image (4:2:0 or 4:4:4), one uninterrupted baseline scan with the tables at d_tables (packed form followed by
// the flat walk's form, as pixo_dev::ScanArgs::tables): pixels -> colour -> DCT -> quantise -> encode_block -> BitWriterMsb's
// bytes (0x00 stuffed behind every 0xFF, the last byte 1-padded) at d_out, in ONE kernel: no coefficient tuple, no packed bit
// stream in HBM.  Replaces launch_jpeg_coeffs + launch_scan_code + launch_stuff_fused for such a scan and leaves their totals:
// host_totals[0] / d_state[1] = the scan's length in bits (unpadded), host_totals[1] = the scan's bytes (stuffed),
// host_totals[2] = its bytes before stuffing, host_totals[3] / d_state[0] = the abort flag of the bounded waits.  Nothing is
// stored beyond d_out[out_cap) — d_out may be device memory or host memory the GPU can write; the caller compares host_totals[1]
// with out_cap and repeats with more room.  d_state: pixels_code_state_words(pixels_code_groups()) u64, ZERO (the launcher clears
// it unless state_is_zero); d_clear / clear_words: words this launch zeroes on the side — the state of the launch BEFORE it,
// which alternates with this one's (a context keeps two, so that no memset launch is needed between files).
// seed_dc: DC predictors of the first Y / Cb / Cr block (null: zeros).
bool pixels_code_supported(uint32_t W, uint32_t H, bool gray);
uint64_t pixels_code_groups(uint32_t W, uint32_t H, bool s420); // workgroups = 512-pixel-wide tiles
size_t pixels_code_state_words(uint64_t groups);
hipError_t launch_pixels_code(const void *d_px, uint32_t W, uint32_t H, bool s420, const float *d_qt, const uint32_t *d_tables,
                              unsigned long long *d_state, bool state_is_zero, unsigned long long *d_clear, size_t clear_words, uint8_t *d_out,
                              uint64_t out_cap, unsigned long long *host_totals, const int16_t seed_dc[3], bool pad_last, void *d_block_spill, hipStream_t s,
                              uint32_t spin_budget = 1u << 20);
// d_block_spill: pixels_code_groups() x 192 x 128 bytes of device memory (never read by the caller): groups whose bits do not fit one
// 6 KiB round (noise at q >= 90) park their quantised blocks there between the rounds' walks

} // namespace pixo_dev
