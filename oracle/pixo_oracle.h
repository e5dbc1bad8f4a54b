/*
 * pixo_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's (leerob/pixo v0.4.1) baseline JPEG
 * encode path, used only as the checker in tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg.  Nothing under pixo_amd/ may include, link or
 * call this.
 *
 * Parity status: PINNED.  Every function here is validated byte-for-byte
 * against the reference's own compiled WebAssembly build (oracle/_ref/pixo_bg.wasm,
 * sha256 832d2c39…0be973) over the matrix in tests/golden/ (see
 * tests/golden/make_golden.py and tests/test_oracle_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 * All floating-point is IEEE binary32, one rounding per operation, no FMA,
 * exactly like rustc/LLVM emits for the reference.
 */
#ifndef PIXO_ORACLE_H
#define PIXO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ColorType discriminants: reference src/color.rs:7-18 (#[repr(u8)]). */
enum { PO_GRAY = 0, PO_GRAY_ALPHA = 1, PO_RGB = 2, PO_RGBA = 3 };
/* Subsampling: reference src/jpeg/mod.rs:96-101. */
enum { PO_S444 = 0, PO_S420 = 1 };

/* Error codes; messages mirror src/error.rs:50-91 Display strings. */
enum {
    PO_OK = 0,
    PO_ERR_INVALID_QUALITY = -1,
    PO_ERR_INVALID_RESTART = -2,
    PO_ERR_INVALID_DIMENSIONS = -3,
    PO_ERR_IMAGE_TOO_LARGE = -4,
    PO_ERR_UNSUPPORTED_COLOR = -5,
    PO_ERR_INVALID_DATA_LENGTH = -6,
    PO_ERR_UNSUPPORTED_OPTION = -7, /* progressive / trellis: outside the restated path */
    PO_ERR_NOMEM = -8
};

typedef struct {
    uint32_t width, height;
    uint8_t color_type;        /* PO_GRAY or PO_RGB */
    uint8_t quality;           /* 1..100 */
    uint8_t subsampling;       /* PO_S444 / PO_S420 */
    uint8_t has_restart;       /* Option<u16> discriminant */
    uint16_t restart_interval;
    uint8_t optimize_huffman;
    uint8_t progressive;       /* not restated: returns PO_ERR_UNSUPPORTED_OPTION */
    uint8_t trellis_quant;     /* not restated: returns PO_ERR_UNSUPPORTED_OPTION */
} po_options;

/* color.rs:60-77 */
void po_rgb_to_ycbcr(uint8_t r, uint8_t g, uint8_t b, uint8_t out[3]);

/* quantize.rs:42-89: zig-zag u8 tables for DQT and natural-order f32 tables. */
void po_quant_tables(uint8_t quality, uint8_t lum_zz[64], uint8_t chr_zz[64],
                     float lum_nat[64], float chr_nat[64]);

/* dct.rs:614-700 (f32 AAN, rows then columns, scale inside each 1-D pass). */
void po_dct_2d(const float in[64], float out[64]);
/* quantize.rs:99-105 */
void po_quantize_block(const float dct[64], const float q[64], int16_t out[64]);
/* quantize.rs:107-113 */
void po_zigzag(const int16_t in[64], int16_t out[64]);
extern const uint8_t PO_ZIGZAG[64];

/* Geometry of the coefficient tuple (jpeg/mod.rs:58-61,1048-1135 layout):
 *   Gray      : y_blocks = ceil(w/8)*ceil(h/8), c_blocks = 0
 *   RGB 4:4:4 : y_blocks = c_blocks = ceil(w/8)*ceil(h/8)      (raster block order)
 *   RGB 4:2:0 : mcus = ceil(w/16)*ceil(h/16); y_blocks = 4*mcus (TL,TR,BL,BR per MCU),
 *               c_blocks = mcus                                  (raster MCU order)
 */
void po_coeff_geometry(uint32_t w, uint32_t h, uint8_t color_type, uint8_t subsampling,
                       size_t *y_blocks, size_t *c_blocks);

/* The pixel pipeline a4..a8 of SURVEY.md §8: colour, extract (edge replicate),
 * 2x2 box, level shift, DCT, quantise.  Natural-order i16[64] per block.
 * threads<=1: single thread; otherwise OpenMP over MCU rows (what the reference's
 * rayon path does in compute_all_coefficients, jpeg/mod.rs:1137-1230). */
/* src/jpeg/trellis.rs:67-208 with DEFAULT_LAMBDA, one block (natural order in and out) */
void po_trellis_quantize(const float dct[64], const float q[64], int16_t out[64]);
/* same with the quantiser of the progressive path: use_trellis != 0 -> trellis_quantize
 * (src/jpeg/trellis.rs:67-208, lambda 1.0) instead of quantize_block */
int po_jpeg_coeffs_ex(const uint8_t *pixels, uint32_t w, uint32_t h, uint8_t color_type,
                      uint8_t subsampling, uint8_t quality, int16_t *y, int16_t *cb, int16_t *cr,
                      int threads, int use_trellis);
int po_jpeg_coeffs(const uint8_t *pixels, uint32_t w, uint32_t h, uint8_t color_type,
                   uint8_t subsampling, uint8_t quality, int16_t *y, int16_t *cb,
                   int16_t *cr, int threads);

/* Whole-file encode: jpeg/mod.rs:328-447 (validation order included) +
 * encode_scan :1408-1563 + huffman.rs:423-481 + bits.rs:195-293.
 * On success *out is malloc'd (caller frees with po_free) and *out_len set. */
int po_encode_jpeg(const uint8_t *data, size_t data_len, const po_options *opt,
                   uint8_t **out, size_t *out_len);

/* Entropy stage only, from an existing coefficient tuple (lets tests feed GPU
 * coefficients through the oracle's bit writer and vice versa). */
int po_encode_jpeg_from_coeffs(const int16_t *y, const int16_t *cb, const int16_t *cr,
                               const po_options *opt, uint8_t **out, size_t *out_len);

/* wasm.rs:113-142 shape: builder order quality -> preset -> subsampling. */
int po_encode_jpeg_flat(const uint8_t *data, size_t data_len, uint32_t w, uint32_t h,
                        uint8_t color_type, uint8_t quality, uint8_t preset,
                        int subsampling_420, uint8_t **out, size_t *out_len);

/* Symbol statistics of a15/a16 (count_block, jpeg/mod.rs:826-860): histograms
 * dc[2][12], ac[2][256] (index 0 = luminance class, 1 = chrominance class). */
int po_symbol_histograms(const int16_t *y, const int16_t *cb, const int16_t *cr,
                         const po_options *opt, uint64_t dc[2][12], uint64_t ac[2][256]);

/* huffman.rs:294-391: BITS/VALS from counts; returns 0 if None (empty or >16 bits). */
int po_build_bits_vals(const uint64_t *counts, int n, uint8_t bits[16], uint8_t *vals,
                       int *nvals);

/* ---- the integer DCT family (pixo_int_oracle.c; dead code in the reference: parity unpinned beyond its unit tests) */
void po_dct_2d_integer(const int16_t block[64], int32_t out[64]);
void po_dct_2d_fast(const int16_t block[64], int32_t out[64]);
void po_quantize_block_integer(const int32_t dct[64], const uint16_t q[64], int16_t out[64]);
void po_rgb_to_ycbcr_2p16(uint8_t r, uint8_t g, uint8_t b, int32_t out[3]);
void po_quant_tables_int(uint8_t quality, uint16_t lum[64], uint16_t chr[64]);
int po_jpeg_coeffs_integer(const uint8_t *pixels, uint32_t w, uint32_t h, uint8_t color_type, uint8_t quality,
                           int16_t *y, int16_t *cb, int16_t *cr);

void po_free(void *p);
const char *po_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif
