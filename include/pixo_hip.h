/*
 * pixo_hip.h — C ABI of the MI355X (gfx950) backend for pixo's JPEG encode path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * A Rust `pixo` crate binds these with `extern "C"` and keeps
 * its public `pixo::jpeg::{encode, encode_into, JpegOptions, …}` surface unchanged.
 * Each entry point cites the reference interface (leerob/pixo v0.4.1) it replaces.
 *
 * Threading: every function is re-entrant; device buffers and the HIP stream live
 * in a thread-local context, so many host threads may encode concurrently (the
 * reference's functions are `Sync`-safe and have rayon callers).
 *
 * Errors: functions return PIXO_OK (0) or a negative pixo_status.  The message of
 * the last failure on the calling thread — identical to the reference's
 * `pixo::Error` Display string (src/error.rs:50-91) — is available from
 * pixo_hip_last_error().  All validation errors are reported before any work is
 * done and in the reference's order (src/jpeg/mod.rs:333-373).
 */
#ifndef PIXO_HIP_H
#define PIXO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pixo::ColorType discriminants (src/color.rs:7-18, #[repr(u8)]). */
enum { PIXO_GRAY = 0, PIXO_GRAY_ALPHA = 1, PIXO_RGB = 2, PIXO_RGBA = 3 };
/* pixo::jpeg::Subsampling (src/jpeg/mod.rs:96-101). */
enum { PIXO_S444 = 0, PIXO_S420 = 1 };

/* One code per pixo::Error variant that this path can raise (src/error.rs:10-48). */
typedef enum {
    PIXO_OK = 0,
    PIXO_ERR_INVALID_DIMENSIONS = -1,      /* Error::InvalidDimensions            */
    PIXO_ERR_INVALID_DATA_LENGTH = -2,     /* Error::InvalidDataLength            */
    PIXO_ERR_INVALID_QUALITY = -3,         /* Error::InvalidQuality               */
    PIXO_ERR_IMAGE_TOO_LARGE = -4,         /* Error::ImageTooLarge                */
    PIXO_ERR_UNSUPPORTED_COLOR_TYPE = -5,  /* Error::UnsupportedColorType         */
    PIXO_ERR_COMPRESSION = -6,             /* Error::CompressionError(String): device/runtime failures */
    PIXO_ERR_INVALID_RESTART_INTERVAL = -7,/* Error::InvalidRestartInterval       */
    PIXO_ERR_INVALID_COLOR_ARG = -8,       /* wasm.rs:122-131 "Invalid color type for JPEG: …" */
    PIXO_ERR_BUFFER_TOO_SMALL = -9         /* encode_into with a fixed-capacity buffer */
} pixo_status;

/* pixo::jpeg::JpegOptions (src/jpeg/mod.rs:121-140), field for field. */
typedef struct pixo_jpeg_options {
    uint32_t width;
    uint32_t height;
    uint8_t color_type;            /* PIXO_GRAY or PIXO_RGB                        */
    uint8_t quality;               /* 1..=100                                      */
    uint8_t subsampling;           /* PIXO_S444 / PIXO_S420                        */
    uint8_t has_restart_interval;  /* Option<u16> discriminant                     */
    uint16_t restart_interval;     /* MCUs; Some(0) is InvalidRestartInterval      */
    uint8_t optimize_huffman;
    uint8_t progressive;           /* SOF2 + simple_progressive_script (7 scans)   */
    uint8_t trellis_quant;         /* acts only with progressive, as upstream      */
} pixo_jpeg_options;

/* JpegOptions::{fast,balanced,max,from_preset} (src/jpeg/mod.rs:162-216). */
void pixo_jpeg_options_from_preset(pixo_jpeg_options *out, uint32_t width, uint32_t height,
                                   uint8_t quality, uint8_t preset);

/* ---- whole-file encode (host pixels in, JFIF bytes out) ------------------------- */

/* Replaces `pixo::jpeg::encode(data, &options) -> Result<Vec<u8>>` (src/jpeg/mod.rs:88).
 * On success *out is a malloc'd buffer the caller releases with pixo_hip_free(). */
int pixo_hip_jpeg_encode(const uint8_t *data, size_t data_len, const pixo_jpeg_options *options,
                         uint8_t **out, size_t *out_len);

/* Replaces `pixo::jpeg::encode_into(&mut output, data, &options)` (src/jpeg/mod.rs:328):
 * writes into caller storage of `capacity` bytes; *out_len receives the bytes needed
 * (also on PIXO_ERR_BUFFER_TOO_SMALL, so a Rust Vec can reserve and retry). */
int pixo_hip_jpeg_encode_into(uint8_t *output, size_t capacity, const uint8_t *data,
                              size_t data_len, const pixo_jpeg_options *options,
                              size_t *out_len);

/* Replaces the flat wasm-bindgen export `encode_jpeg(data,width,height,color_type,
 * quality,preset,subsampling_420)` (src/wasm.rs:113-142): same 7 arguments, same
 * builder order (.quality → .preset → .subsampling), same error strings. */
int pixo_hip_encode_jpeg(const uint8_t *data, size_t data_len, uint32_t width, uint32_t height,
                         uint8_t color_type, uint8_t quality, uint8_t preset,
                         int subsampling_420, uint8_t **out, size_t *out_len);

/* ---- the device seam: the coefficient tuple ------------------------------------- */

/* Geometry of `YCbCrCoefficients` (src/jpeg/mod.rs:58-61, filled by
 * compute_all_coefficients :932-1230): natural-order i16[64] blocks; 4:2:0 stores
 * Y as 4 blocks per MCU (TL,TR,BL,BR) in raster MCU order. */
int pixo_hip_coeff_geometry(uint32_t width, uint32_t height, uint8_t color_type,
                            uint8_t subsampling, size_t *y_blocks, size_t *c_blocks);

/* Fused colour → (2x2 box) → level shift → f32 AAN DCT → quantise on the GPU;
 * replaces the per-MCU loop bodies of encode_scan / compute_all_coefficients
 * (src/jpeg/mod.rs:1448-1557, :981-1046: extract_block/extract_mcu_420 + dct_2d +
 * quantize_block).  Host pointers; H2D/D2H copies included; synchronous. */
int pixo_hip_jpeg_coeffs(const uint8_t *pixels, uint32_t width, uint32_t height,
                         uint8_t color_type, uint8_t subsampling, uint8_t quality,
                         int16_t *y, size_t y_blocks, int16_t *cb, int16_t *cr,
                         size_t c_blocks);

/* Same computation on DEVICE pointers for `batch` equally-sized images laid out
 * back to back (pixels: batch*w*h*bpp bytes; y: batch*y_blocks*64 i16; …).
 * Asynchronous on `stream` (a hipStream_t, NULL = default stream). */
int pixo_hip_jpeg_coeffs_device(const void *d_pixels, uint32_t width, uint32_t height,
                                uint8_t color_type, uint8_t subsampling, uint8_t quality,
                                uint32_t batch, void *d_y, void *d_cb, void *d_cr,
                                void *stream);

/* The INTEGER secondary mode of the coefficient stage — never the default: `pixo::jpeg::encode`
 * runs the f32 transform above, and only that path gives the reference's bytes.  This entry exposes the fixed-point
 * family the reference also carries (dead code upstream): colour by the 2^16 row formulas of rgb_to_ycbcr_row_avx2
 * (src/simd/x86_64.rs:1330-1420), dct_2d_fast / dct_2d_integer (src/jpeg/dct.rs:535-568, :61-186: 13-bit fixed point,
 * per-product truncation, constant-block shortcut), quantize_block_integer (src/jpeg/dct.rs:570-583) with the
 * `*_table_int` tables (src/jpeg/quantize.rs:56-78), per 8x8 block with extract_block's edge replication
 * (src/jpeg/mod.rs:1565-1606).  4:4:4 RGB or gray (subsampling PIXO_S420 is refused); same tuple layout as above.
 * Host pointers, synchronous / device pointers of the current device, asynchronous on `stream`. */
int pixo_hip_jpeg_coeffs_integer(const uint8_t *pixels, uint32_t width, uint32_t height, uint8_t color_type,
                                 uint8_t subsampling, uint8_t quality, int16_t *y, size_t y_blocks, int16_t *cb,
                                 int16_t *cr, size_t c_blocks);
int pixo_hip_jpeg_coeffs_integer_device(const void *d_pixels, uint32_t width, uint32_t height, uint8_t color_type,
                                        uint8_t subsampling, uint8_t quality, void *d_y, void *d_cb, void *d_cr,
                                        void *stream);

/* Entropy stage on the host from a coefficient tuple (zig-zag, DC delta, run-length,
 * Huffman, byte stuffing, headers): replaces encode_block + BitWriterMsb + write_*
 * (src/jpeg/huffman.rs:423-481, src/bits.rs:195-293, src/jpeg/mod.rs:449-648) when
 * the coefficients were produced elsewhere (e.g. by several GPUs, see bands below). */
int pixo_hip_jpeg_entropy_encode(const int16_t *y, const int16_t *cb, const int16_t *cr,
                                 const pixo_jpeg_options *options, uint8_t **out,
                                 size_t *out_len);

/* Entropy stage on the DEVICE from a coefficient tuple that is already in HBM (as produced by
 * pixo_hip_jpeg_coeffs_device for ONE image): per-block Huffman coding, bit-offset prefix sum,
 * parallel packing and 0xFF stuffing on the GPU, byte-identical to encode_scan + encode_block +
 * BitWriterMsb (src/jpeg/mod.rs:1408-1563, src/jpeg/huffman.rs:423-481, src/bits.rs:195-293);
 * with options->optimize_huffman the count_block statistics (src/jpeg/mod.rs:826-860) are
 * gathered on the GPU as well, and restart intervals (src/jpeg/mod.rs:1423-1445) are handled as
 * byte-aligned segments with RSTn markers.  Only the finished file crosses PCIe.  Synchronous;
 * the pointers must belong to the current HIP device; work enqueued on the producer stream
 * (pixo_hip_set_producer_stream) is waited for on the device.  The tuple is coded as it is:
 * options->trellis_quant with progressive is refused (the tuple entries cannot re-quantise). */
int pixo_hip_jpeg_entropy_encode_device(const void *d_y, const void *d_cb, const void *d_cr,
                                        const pixo_jpeg_options *options, uint8_t **out,
                                        size_t *out_len);

/* pixo::jpeg::encode for pixels that are already in HBM (options->width * height * bpp bytes,
 * tightly packed): coefficient kernel + device entropy stage, result in malloc'd host memory.
 * Same validation and errors as pixo_hip_jpeg_encode.  Ordered after the producer stream (see
 * pixo_hip_set_producer_stream), like every entry point below that takes device pixels. */
int pixo_hip_jpeg_encode_device(const void *d_pixels, const pixo_jpeg_options *options,
                                uint8_t **out, size_t *out_len);

/* The same into caller storage (`encode_into` for resident pixels): headers and the entropy-coded bytes
 * are written straight into `output` — with pinned (hipHostMalloc / registered) storage the device-to-host
 * copy is the only pass over the file.  *out_len receives the file size, also on
 * PIXO_ERR_BUFFER_TOO_SMALL (what `output` holds then is unspecified: a pinned `output` is written by the GPU directly,
 * up to its capacity).  A large scan (more than 786,432 blocks: 8192 x 4096 at
 * 4:2:0) is coded in pieces whose bytes travel while the next piece is coded — into `output` directly when
 * capacity >= 64 bytes per 8x8 block + 10 KB (a smaller `output` gets the file in one copy at the end; an `output` that
 * large may have been written to when a file larger still is refused with PIXO_ERR_BUFFER_TOO_SMALL). */
int pixo_hip_jpeg_encode_device_into(const void *d_pixels, const pixo_jpeg_options *options,
                                     uint8_t *output, size_t capacity, size_t *out_len);

/* pixo::jpeg::encode for `batch` equally sized images back to back in HBM (config 3: 64 x 1080p):
 * one coefficient launch and ONE pass of the device entropy stage for all of them — every image a byte-aligned
 * segment of the two single-pass entropy kernels — files[i] / lens[i] receive `batch` malloc'd files
 * (pixo_hip_free).  With optimize_huffman, progressive scans or restart markers the images are encoded one by one. */
int pixo_hip_jpeg_encode_batch_device(const void *d_pixels, const pixo_jpeg_options *options, uint32_t batch,
                                      uint8_t **files, size_t *lens);

/* The same into ONE block of caller storage: file i occupies arena[offsets[i], offsets[i] + lens[i]), the files follow
 * each other without gaps (offsets[0] = 0); every file is copied from the device straight to its final place — with
 * pinned (hipHostMalloc / registered) storage the device-to-host copy is the only pass over the bytes.  offsets and lens
 * have `batch` entries.  When the files do not fit, offsets / lens are still filled in (capacity needed =
 * offsets[batch - 1] + lens[batch - 1]) and PIXO_ERR_BUFFER_TOO_SMALL is returned; a null arena with capacity 0 is a size
 * query (nothing is copied).  Batches of 64 MB of pixels and more are coded in sub-batches whose files cross PCIe while the
 * next sub-batch's kernels run (two of the library's contexts alternate): an arena that turns out too small may then
 * hold the files of the first sub-batches.  `arena` may also be DEVICE memory (hipMalloc, on the current device): the
 * files then stay in HBM, complete with headers and EOI, for a caller that moves them on itself — pixo_amd/sharded.py
 * gathers the files of a batch that was scattered over several GPUs with RCCL ("C3 batch").
 * (A batch that is NOT coded in one pass — optimize_huffman, progressive scans, restart markers — reaches a device arena image by
 * image through host files: every image crosses PCIe twice there; a too small arena is noticed only after all images were coded.)
 * Replaces a loop over pixo::jpeg::encode_into (src/jpeg/mod.rs:328). */
int pixo_hip_jpeg_encode_batch_device_into(const void *d_pixels, const pixo_jpeg_options *options, uint32_t batch,
                                           uint8_t *arena, size_t capacity, size_t *offsets, size_t *lens);

/* ---- PNG row filters + Adler-32 ------------------------------ */

/* pixo::png::FilterStrategy in declaration order (src/png/mod.rs:345-364); every strategy runs on the
 * device. */
enum pixo_png_filter_strategy {
    PIXO_PNG_NONE = 0, PIXO_PNG_SUB = 1, PIXO_PNG_UP = 2, PIXO_PNG_AVERAGE = 3, PIXO_PNG_PAETH = 4,
    PIXO_PNG_MINSUM = 5, PIXO_PNG_ADAPTIVE = 6, PIXO_PNG_ADAPTIVE_FAST = 7, PIXO_PNG_BIGRAMS = 8
};
/* flags */
#define PIXO_PNG_NO_RAYON 1u /* semantics of a build without the `parallel` feature (wasm): AdaptiveFast is
                                 always the sequential, stateful variant (src/png/filter.rs:147-167) */

/* Replaces pixo::png::filter::apply_filters (src/png/filter.rs:51-206) and the Adler-32 the zlib
 * wrapper computes over its result (src/simd/fallback.rs:8-25, src/compress/deflate.rs:1044):
 * `out` receives height * (width * bytes_per_pixel + 1) bytes — filter-type byte + filtered row,
 * exactly what the reference hands to its DEFLATE — and *adler32 the checksum of those bytes.
 * Same strategy rules as the reference: images of <= 4096 pixels use Sub instead of the adaptive
 * strategies; the adaptive strategies treat rows independently when height > 32 (the rayon path
 * of the default build) and sequentially otherwise.  bytes_per_pixel in {1,2,3,4,6,8}.
 * Host pointers; synchronous. */
int pixo_hip_png_filter(const uint8_t *data, size_t data_len, uint32_t width, uint32_t height,
                        uint32_t bytes_per_pixel, uint8_t strategy, uint32_t flags, uint8_t *out,
                        size_t out_capacity, uint32_t *adler32);

/* Same on DEVICE pointers of the current HIP device (d_out: height * (row bytes + 1) bytes).
 * Synchronous: returns after the checksum has been combined. */
int pixo_hip_png_filter_device(const void *d_data, uint32_t width, uint32_t height, uint32_t bytes_per_pixel,
                               uint8_t strategy, uint32_t flags, void *d_out, uint32_t *adler32);

/* Asynchronous form for resident pipelines: enqueues the filter kernel on `stream` (a
 * hipStream_t, NULL = default stream) and leaves the per-row checksum partials — two u64 per row:
 * byte sum, position-weighted byte sum — in d_row_sums[2 * height]; d_scratch is 16 bytes of
 * device memory.  pixo_hip_png_adler32_from_row_sums turns a HOST copy of the partials into the
 * zlib Adler-32 of the whole filtered stream. */
int pixo_hip_png_filter_async(const void *d_data, uint32_t width, uint32_t height, uint32_t bytes_per_pixel,
                              uint8_t strategy, uint32_t flags, void *d_out, void *d_row_sums,
                              void *d_scratch, void *stream);
uint32_t pixo_hip_png_adler32_from_row_sums(const uint64_t *row_sums, uint32_t width, uint32_t height,
                                            uint32_t bytes_per_pixel);

/* ---- multi-GPU band sharding -------------------------------------- */

/* Splits the image into `parts` contiguous MCU-row bands; band `index` covers pixel
 * rows [*row_begin, *row_end) and owns *y_blocks / *c_blocks of the tuple starting
 * at block offsets *y_offset / *c_offset.  Concatenating the bands' coefficients in
 * index order gives exactly the single-device tuple (edge replication included). */
int pixo_hip_band(uint32_t width, uint32_t height, uint8_t color_type, uint8_t subsampling,
                  uint32_t parts, uint32_t index, uint32_t *row_begin, uint32_t *row_end,
                  size_t *y_offset, size_t *y_blocks, size_t *c_offset, size_t *c_blocks);

/* ---- one image across several GPUs: per-band entropy coding + splice ------------ */

/* The reference has no device boundary; its seam for this is the scan loop's only cross-MCU state: the
 * three DC predictors and the bit writer's position (src/jpeg/mod.rs:1417-1419, src/bits.rs:216-272).
 * A band (pixo_hip_band) is coded on its own GPU given (1) the last DCs of the band above and (2) the
 * bit offset at which it starts in the scan; what the GPUs exchange is 3 x i16 and one u64 per band
 * (plus, for optimised tables, 536 counters per band, summed) — never coefficients.  The exchange is the
 * caller's: torch.distributed all_gather between processes (pixo_amd/sharded.py), shared memory between
 * threads (pixo_hip_jpeg_encode_multi below).  Baseline scans without restart markers; other option sets
 * are refused by _create (gather the coefficient bands and use pixo_hip_jpeg_entropy_encode_device).
 * Calls on one encoder must not overlap; different encoders may run on different threads at once. */
#define PIXO_HIP_COUNT_WORDS 536 /* [class 0 luminance, 1 chrominance][12 DC categories + 256 AC run/size symbols] */
typedef struct pixo_hip_band_encoder pixo_hip_band_encoder;

/* Band `index` of `parts` of the image `options` describes, on HIP device `device`. */
int pixo_hip_band_encoder_create(const pixo_jpeg_options *options, uint32_t parts, uint32_t index, int device,
                                 pixo_hip_band_encoder **out);
void pixo_hip_band_encoder_destroy(pixo_hip_band_encoder *encoder);
/* pixel rows [*row_begin, *row_end) of the image belong to this band (equal: more bands than MCU rows) */
int pixo_hip_band_encoder_rows(const pixo_hip_band_encoder *encoder, uint32_t *row_begin, uint32_t *row_end);
/* Step 1: the coefficient kernel over the band's rows (`band_pixels`: those rows only, tightly packed; a
 * host pointer — copied over this GPU's own PCIe link — or, with on_device != 0, a pointer on the
 * encoder's device).  last_dc: quantised DC of the band's last Y, Cb, Cr block — what the NEXT band's
 * first blocks predict from (encode_block's return value, src/jpeg/huffman.rs:480).  A band without
 * rows reports zeros; its successor takes the DCs of the nearest band above that has rows. */
int pixo_hip_band_encoder_coeffs(pixo_hip_band_encoder *encoder, const void *band_pixels, int on_device,
                                 int16_t last_dc[3]);
/* Step 2, only with optimize_huffman: the band's count_block statistics (src/jpeg/mod.rs:826-860) given its
 * predecessors' DCs.  The tables are built from the element-wise SUM over all bands. */
int pixo_hip_band_encoder_count(pixo_hip_band_encoder *encoder, const int16_t prev_dc[3],
                                uint64_t counts[PIXO_HIP_COUNT_WORDS]);
/* Step 3: the band's length in bits (total_counts: the summed statistics, NULL without optimize_huffman). */
int pixo_hip_band_encoder_lengths(pixo_hip_band_encoder *encoder, const int16_t prev_dc[3],
                                  const uint64_t *total_counts, uint64_t *bits);
/* Step 4: Huffman-code, pack and 0xFF-stuff the band at `bit_offset` (the sum of the bits of all bands
 * before it) -> *piece (pixo_hip_free): 16 header bytes { head_nbits, head_bits, tail_nbits, tail_bits,
 * 0,0,0,0, body_len u64 LE } + body.  head = the band's first (8 - bit_offset % 8) % 8 bits (they share a
 * byte with the band before), body = its whole bytes, stuffed, tail = the bits left over. */
int pixo_hip_band_encoder_pack(pixo_hip_band_encoder *encoder, uint64_t bit_offset, uint8_t **piece,
                               size_t *piece_len);
/* Step 4 without the copy: the stuffed body stays in the encoder's device buffer (*d_body, *body_len; valid
 * until the encoder's next call; both may be NULL) and only the 16 header bytes come back.  _copy_body then
 * moves it into caller storage — typically its final place in the file (pixo_hip_jpeg_splice_layout), each
 * band over its own GPU's PCIe link; registered / pinned storage is written directly, and `dst` may also be
 * device memory (the send buffer of a collective). */
int pixo_hip_band_encoder_pack_device(pixo_hip_band_encoder *encoder, uint64_t bit_offset, uint8_t header[16],
                                      void **d_body, size_t *body_len);
int pixo_hip_band_encoder_copy_body(pixo_hip_band_encoder *encoder, uint8_t *dst);
/* Host: headers (src/jpeg/mod.rs:449-648) + the pieces of all bands in order, shared bytes merged and
 * stuffed, final 1-padding (BitWriterMsb::flush) + EOI = the file pixo::jpeg::encode writes. */
int pixo_hip_jpeg_splice(const pixo_jpeg_options *options, const uint64_t *total_counts,
                         const uint8_t *const *pieces, const size_t *piece_lens, uint32_t parts,
                         uint8_t **out, size_t *out_len);
/* The same in two steps, so that no body is copied twice: from the 16-byte headers of all pieces
 * (`piece_headers`: parts x 16 bytes) _layout tells the file's length and where every body belongs;
 * _finish writes everything else into `file` — JFIF headers, the bytes neighbouring bands share (merged and
 * stuffed), the final 1-padding, EOI. */
int pixo_hip_jpeg_splice_layout(const pixo_jpeg_options *options, const uint64_t *total_counts,
                                const uint8_t *piece_headers, uint32_t parts, size_t *file_len,
                                size_t *body_offsets);
int pixo_hip_jpeg_splice_finish(const pixo_jpeg_options *options, const uint64_t *total_counts,
                                const uint8_t *piece_headers, uint32_t parts, uint8_t *file, size_t file_len);
/* Host twins of steps 2-4 for a band whose coefficient tuple is in host memory (`band_rows` pixel rows;
 * what pixo_hip_jpeg_entropy_encode is to the whole tuple). */
int pixo_hip_jpeg_band_count_host(const int16_t *y, const int16_t *cb, const int16_t *cr,
                                  const pixo_jpeg_options *options, uint32_t band_rows, const int16_t prev_dc[3],
                                  uint64_t counts[PIXO_HIP_COUNT_WORDS]);
int pixo_hip_jpeg_band_bits_host(const int16_t *y, const int16_t *cb, const int16_t *cr,
                                 const pixo_jpeg_options *options, uint32_t band_rows, const int16_t prev_dc[3],
                                 const uint64_t *total_counts, uint64_t *bits);
int pixo_hip_jpeg_band_piece_host(const int16_t *y, const int16_t *cb, const int16_t *cr,
                                  const pixo_jpeg_options *options, uint32_t band_rows, const int16_t prev_dc[3],
                                  const uint64_t *total_counts, uint64_t bit_offset, uint8_t **piece,
                                  size_t *piece_len);

/* pixo::jpeg::encode on `n_devices` GPUs of this process (config 4: one 16384 x 16384 image over 8
 * MI355X): band k of n_devices goes to devices[k] (a device may appear more than once); every band's
 * rows are read straight from `data` over that GPU's PCIe link by its own host thread, the exchanges
 * above happen in shared memory, the calling thread splices.  Byte-identical to pixo_hip_jpeg_encode.
 * Progressive scans and restart markers are coded by devices[0] alone. */
int pixo_hip_jpeg_encode_multi(const uint8_t *data, size_t data_len, const pixo_jpeg_options *options,
                               const int *devices, uint32_t n_devices, uint8_t **out, size_t *out_len);

/* A BATCH over `n_devices` GPUs of this process (configs[2] on a node, SURVEY §8e "C3 batch"): `batch` equally sized images
 * back to back at `pixels` — memory of ANY of the process's GPUs (the images of the other GPUs' shares travel by peer copies,
 * hipMemcpyPeer: one peer per xGMI link) or host memory (every GPU fetches its share over its own PCIe link).  devices[k]
 * encodes images [batch k / n, batch (k + 1) / n) with pixo_hip_jpeg_encode_batch_device_into's one-pass kernels into its own
 * HBM, the persistent worker threads of pixo_hip_jpeg_encode_multi exchange the file lengths in shared memory, and every GPU
 * copies its run of finished files to their final place in `arena` (host memory, pinned for full speed) over its OWN PCIe
 * link.  offsets / lens / capacity / PIXO_ERR_BUFFER_TOO_SMALL as pixo_hip_jpeg_encode_batch_device_into (a null arena with
 * capacity 0 is a size query); a device may appear more than once.  Byte-identical to that entry on one GPU.
 * Device pixels: what the caller enqueued on its producer stream (pixo_hip_set_producer_stream; default the NULL stream) is waited
 * for before the workers read them — `pixels` needs no length argument, so the caller guarantees batch x image bytes behind it
 * (the Python and Rust mirrors check that).  pixo_hip_trim() also releases the workers' device buffers.
 * Replaces a loop over pixo::jpeg::encode (src/jpeg/mod.rs:88) spread over the GPUs of a node. */
int pixo_hip_jpeg_encode_batch_multi(const void *pixels, const pixo_jpeg_options *options, uint32_t batch,
                                     const int *devices, uint32_t n_devices, uint8_t *arena, size_t capacity,
                                     size_t *offsets, size_t *lens);

/* ---- runtime ----------------------------------------------------------------------- */

int pixo_hip_device_count(void);            /* 0 when no GPU / no driver              */
int pixo_hip_set_device(int device);        /* device for the calling thread's context (host-pointer entries) */
/* Entry points that take DEVICE pointers run on the library's own stream, ordered after everything the
 * caller has enqueued so far on `stream` (a hipStream_t of the calling thread's current device; default:
 * the NULL stream, which is also PyTorch's default stream).  Per thread. */
int pixo_hip_set_producer_stream(void *stream);
void *pixo_hip_get_producer_stream(void);   /* what the calling thread set (to save and restore it around a call) */
/* Releases the calling thread's device and pinned buffers AND every context parked by threads that have ended.
 * Buffers only grow while a thread lives; when it ends its context is parked for the next thread (no hipMalloc per
 * request in a thread-per-request server).  Parked contexts are bounded — at most 16 of them and 8 GiB of device +
 * pinned memory together, the oldest give their large buffers back first — but a LIVE thread keeps what its largest
 * image needed (a 16384x16384 image: ~2 GB of HBM, ~0.4 GB pinned) until it calls this. */
int pixo_hip_trim(void);
/* Tests and A/B tools only: replaces the debug switches read from the environment variable PIXO_HIP_DEBUG
 * ("name[=value],...": trace, host_entropy, multipass_entropy, direct_stores, one_piece, two_kernel_scan, fused_batch, no_side_stats, coef_form=scalar|packed, trellis_form=lane|group, batch_parts=n, piece_groups=n, piece_medium=n,
 * piece_schedule=a:b:c, copy_threads=n, spin_budget=n, no_bands_upload, bands_upload_min_mb=n, bands_upload_mb=n — pixo_amd/csrc/capi_internal.hpp).  None of
 * them changes the bytes of a file.  NULL = read the environment again.  Not synchronised with calls in flight. */
int pixo_hip_debug_configure(const char *switches_or_null);
/* How often a single-pass entropy kernel gave up waiting (its waits on other workgroups are bounded) and the scan was
 * coded again by the multi-pass kernels, in this process.  0 in normal operation. */
uint64_t pixo_hip_debug_lookback_fallbacks(void);
/* The dispatch gate of the single-pass kernels (pixo_amd/csrc/dispatch_gate.hpp: calls of several threads are kept from
 * starting such kernels on an empty device at the same moment — two launches that split the device's workgroup slots wait for
 * each other until the bounded waits give up): how often a launch found the launch before it, another thread's, not yet fully
 * dispatched and waited for it (*waits), and how often it stopped waiting after 5 ms (*timeouts).  Either pointer may be NULL. */
int pixo_hip_debug_dispatch_gate(uint64_t *waits, uint64_t *timeouts);
/* MEASUREMENT only (bench.py, tools/ab_binaries.py): a plain copy of `bytes` bytes of device memory in the coefficient
 * kernel's launch shape — one generation of 192-thread workgroups, 24 KiB each, 8 non-temporal 16-byte loads then 8
 * non-temporal stores per thread, no arithmetic (pixo_amd/csrc/stream_copy.hip) — so that a run can report what the
 * memory system of THIS box gives the kernel's 50 MB + 50 MB beside the kernel's own time.  Replaces nothing of the
 * reference.  `bytes` must be a multiple of 24576, the pointers 16-byte aligned; asynchronous on `stream`. */
int pixo_hip_debug_stream_copy(const void *d_in, void *d_out, size_t bytes, void *stream);
/* MEASUREMENT only: the same for a kernel that writes more or less than it reads — `workgroups` workgroups of 192 threads, every
 * thread `loads` non-temporal 16-byte loads, then `stores` non-temporal 16-byte stores (each in {1, 2, 4, 8, 16}): in = workgroups
 * x loads x 3072 bytes, out = workgroups x stores x 3072 bytes.  The 4:4:4 coefficient kernel is (tiles, 4, 8), the PNG filter
 * kernel (n, 8, 8), the fused pixel -> scan kernel on noise (tiles, 8, 2).  bench.py reports every kernel line beside it
 * (`copy_us_same_run`). */
int pixo_hip_debug_stream_io(const void *d_in, void *d_out, uint32_t workgroups, uint32_t loads, uint32_t stores, void *stream);
/* MEASUREMENT only: the engine clock in Hz while every SIMD issues vector instructions (four wavefronts each, ~1 ms of v_add_f32
 * chains; shader-clock ticks over constant-clock ticks).  The chip clocks down under vector load: bench.py puts a kernel's active
 * vector-ALU cycles over 1,024 SIMDs x THIS clock x the kernel's time (`frac_issue`), not over the 2.4 GHz peak.  Synchronises `stream`. */
int pixo_hip_debug_engine_clock(void *stream, double *hz);
/* MEASUREMENT only: the DEVICE work of one baseline file with standard tables — pixels -> the finished (stuffed, padded) scan in
 * the context's device buffer — enqueued on `stream` and NOT waited for; nothing is delivered.  The kernels are the ones
 * pixo_hip_jpeg_encode_device[_into] runs: the fused pixel -> scan kernel (jpeg_pixels_code.hip: ONE kernel; *form = 1) or
 * coefficient kernel + scan_code + stuffing kernel (*form = 0: gray images, debug switch two_kernel_scan).  K calls between
 * two events on `stream` = the device time per file without the call's waits and the file's way over PCIe.  The caller
 * synchronises the stream before it calls anything else of this library on the same thread. */
int pixo_hip_debug_scan_device_async(const void *d_pixels, const pixo_jpeg_options *options, void *stream, int *form);
/* ... the same for `batch` equally sized images back to back in device memory (configs[2]: 64 x 1920x1080) — the device work of
 * pixo_hip_jpeg_encode_batch_device[_into]'s one-pass form, the scans left in the context's device buffer at their files'
 * spacing: ONE launch of the fused kernel with every image a segment (*form = 1), or coefficient kernel + scan_code + stuffing
 * kernel over the batch (*form = 0).  batch = 1 is the entry above. */
int pixo_hip_debug_scan_device_async_batch(const void *d_pixels, const pixo_jpeg_options *options, uint32_t batch, void *stream, int *form);
/* Releases a buffer the library returned.  Blocks of 24 MiB and more are kept (at most two, 1 GiB) for the next large
 * file instead of going back to the system — their pages are resident, a fresh block of that size costs more than the
 * encode (profiles/r03_fresh_pages.txt); pixo_hip_trim() returns them. */
void pixo_hip_free(void *p);
/* memcpy of a finished file into storage of the caller's, by the library's copy threads (files of 2 MiB and more),
 * with a transparent-huge-page hint for a large destination that has not been touched yet.  For bindings that must
 * hand the file to their runtime in an object of the runtime's own (a Python bytes, a Java byte[]): the copy into a
 * fresh 178 MB object is 28 ms on one thread, 3 ms this way. */
void pixo_hip_copy_file(void *dst, const void *src, size_t n);
const char *pixo_hip_last_error(void);      /* thread-local, never NULL               */
const char *pixo_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PIXO_HIP_H */
