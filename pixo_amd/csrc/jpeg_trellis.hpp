// jpeg_trellis.hpp — launcher of the trellis quantisation kernel (jpeg_trellis.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace pixo_dev {
// d_raw: nblocks x 64 f32 DCT coefficients (natural order) as left by the coefficient kernel's raw
// mode — the planes of a tuple back to back: blocks [0, nluma) are quantised with d_q_luma, the others
// with d_q_chroma (64 steps each, natural order).  d_out: nblocks x 64 i16.  d_scratch: back-pointer
// storage of trellis_scratch_bytes(nblocks) bytes (504 per block), free again when the kernel is done.
size_t trellis_scratch_bytes(uint64_t nblocks);
// Up to this many blocks the search runs on EIGHT lanes per block (a wavefront = 8 blocks; jpeg_trellis.hip), above on one lane
// per block (a wavefront = 64 blocks).  set_trellis_form: 0 = that rule, 1 = always one lane, 2 = always eight (tests, A/B).
constexpr uint64_t kTrellisLanesBlocks = 32768;
void set_trellis_form(int form);
hipError_t launch_trellis(const float *d_raw, const float *d_q_luma, const float *d_q_chroma, int16_t *d_out, uint64_t nblocks,
                          uint64_t nluma, void *d_scratch, hipStream_t s);
} // namespace pixo_dev
