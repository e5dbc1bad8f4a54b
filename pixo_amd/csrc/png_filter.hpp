// png_filter.hpp — host-callable launcher of the PNG row-filter kernel (png_filter.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace pixo_dev {

// FilterStrategy in the reference's declaration order (src/png/mod.rs:345-364)
enum { PNG_S_NONE = 0, PNG_S_SUB, PNG_S_UP, PNG_S_AVERAGE, PNG_S_PAETH, PNG_S_MINSUM, PNG_S_ADAPTIVE, PNG_S_ADAPTIVE_FAST,
       PNG_S_BIGRAMS };

// Filters `height` rows of `width * bpp` bytes (device pointers).  d_out receives height * (width*bpp + 1)
// bytes; d_row_sums [height][2] the per-row Adler partial sums (byte sum, position-weighted sum) the
// caller combines; d_scratch is one int.  sequential_fast selects the stateful AdaptiveFast of builds
// without rayon / of images with height <= 32 (src/png/filter.rs:147-167).
hipError_t launch_png_filter(const void *d_data, uint32_t width, uint32_t height, uint32_t bpp, int strategy,
                             bool sequential_fast, void *d_out, unsigned long long *d_row_sums, int *d_scratch,
                             hipStream_t stream);
// rows [first_row, first_row + rows) only (the row above the first must already be in d_data; not for the stateful
// AdaptiveFast, which needs row 0's decision first)
hipError_t launch_png_filter_rows(const void *d_data, uint32_t width, uint32_t height, uint32_t bpp, int strategy,
                                  bool sequential_fast, void *d_out, unsigned long long *d_row_sums, int *d_scratch,
                                  uint32_t first_row, uint32_t rows, hipStream_t stream);

} // namespace pixo_dev
