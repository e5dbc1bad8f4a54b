// jpeg_pixels_code.hpp — host-callable launcher of the fused pixel -> packed bit stream kernel (jpeg_pixels_code.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace pixo_dev {

// One RGB image (4:2:0 or 4:4:4), one uninterrupted baseline scan with the tables at d_tables (packed form followed by
// the flat walk's form, as pixo_dev::ScanArgs::tables): pixels -> colour -> DCT -> quantise -> encode_block -> the packed
// MSB-first bit stream from bit 0 of d_stream, in ONE kernel; no coefficient tuple is written.  Replaces
// launch_jpeg_coeffs + launch_scan_code for such a scan and leaves what launch_scan_code leaves: d_state[1] / host_totals[0]
// = the scan's length in bits (unpadded), host_totals[3] / d_state[0] = the abort flag of the bounded waits, the words at
// d_clear zeroed for the stuffing launch that follows (launch_stuff_fused with code_state_words = pixels_code_state_words()).
// d_state: pixels_code_state_words(pixels_code_groups()) u64, zero (the launcher clears it unless state_is_zero);
// d_stream: room for blocks * 209 + 64 bytes.  seed_dc: DC predictors of the first Y / Cb / Cr block (null: zeros).
bool pixels_code_supported(uint32_t W, uint32_t H, bool gray);
uint64_t pixels_code_groups(uint32_t W, uint32_t H, bool s420); // workgroups = 512-pixel-wide tiles
size_t pixels_code_state_words(uint64_t groups);
hipError_t launch_pixels_code(const void *d_px, uint32_t W, uint32_t H, bool s420, const float *d_qt, const uint32_t *d_tables,
                              unsigned long long *d_state, bool state_is_zero, uint32_t *d_stream, unsigned long long *d_clear, size_t clear_words,
                              unsigned long long *host_totals, const int16_t seed_dc[3], bool pad_last, hipStream_t s, uint32_t spin_budget = 1u << 20);

} // namespace pixo_dev
