// jpeg_trellis.hpp — launcher of the trellis quantisation kernel (jpeg_trellis.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace pixo_dev {
// d_raw: nblocks x 64 f32 DCT coefficients (natural order) as left by the coefficient kernel's raw
// mode, d_q: 64 quantiser steps (natural order), prescale: 1, or 0.25 for the 4:2:0 chroma blocks whose
// transform ran on 2x2 sums.  d_out: nblocks x 64 i16.
hipError_t launch_trellis(const float *d_raw, const float *d_q, float prescale, int16_t *d_out, uint64_t nblocks, hipStream_t s);
} // namespace pixo_dev
