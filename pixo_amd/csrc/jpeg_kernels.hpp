// jpeg_kernels.hpp — host-callable launcher of the gfx950 coefficient kernel.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace pixo_dev {

// Which form of the DCT passes and the quantiser a launch of `workgroups` tiles takes (jpeg_tile.h, block_rows): one generation
// (all workgroups resident at once: 8 per CU x 256 CUs) is latency-bound at its end and runs the scalar forms; several
// generations are issue-bound and run the packed ones.
bool packed_launch(uint64_t workgroups);
// measurements: 0 = by launch size (above), 1 = always the scalar forms, 2 = always the packed ones (PIXO_HIP_DEBUG coef_form=scalar|packed)
void set_coef_form(int form);

// Enqueues the fused colour -> DCT -> quantise kernel for `batch` equally sized images on
// `stream`.  All pointers are device pointers; d_qt points at the 512-float table block of
// the requested quality (layout in jpeg_tile.h).  d_cb/d_cr are ignored for gray input.
// raw_f32 (batch 1): the UNQUANTISED transform instead (the reference's dct_2d output, f32), input of the trellis quantiser
// (jpeg_trellis.hpp) and in its layout: d_y, d_cb = d_y + 64 y_blocks, d_cr = d_cb + 64 c_blocks are ONE run of blocks,
// coefficient i (natural order) of block B at d_y[(B / 64 * 64 + i) * 64 + B % 64]; room for the run rounded up to 64 blocks.
hipError_t launch_jpeg_coeffs(const void *d_px, uint32_t W, uint32_t H, bool gray, bool s420,
                              uint32_t batch, void *d_y, void *d_cb, void *d_cr,
                              const float *d_qt, hipStream_t stream, bool raw_f32 = false);

// Is the current device a WHOLE MI355X (256 CUs: 2048 of the 192-thread workgroups are one resident generation)?  The late start of
// half a generation (coefficient kernel, PNG filter kernel) is tuned for that and switched off on partitioned devices.
bool whole_chip_device();

// host_out[0..2] (pinned host memory) = the quantised DC of the last block of the Y / Cb / Cr plane (0 for planes without
// blocks): what the next band of an image spread over several GPUs predicts from.  One launch, no copy.
hipError_t launch_last_dcs(const void *d_y, size_t y_blocks, const void *d_cb, const void *d_cr, size_t c_blocks, int16_t *host_out, hipStream_t stream);

// The INTEGER secondary mode (jpeg_integer.hip): 4:4:4 RGB or gray, one image; ql / qc = the natural-order
// integer quantiser tables of the requested quality (host memory, passed by value to the kernel).
hipError_t launch_jpeg_coeffs_integer(const void *d_px, uint32_t W, uint32_t H, bool gray, const uint16_t ql[64], const uint16_t qc[64],
                                      void *d_y, void *d_cb, void *d_cr, hipStream_t stream);

} // namespace pixo_dev
