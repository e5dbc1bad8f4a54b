// jpeg_host.hpp — host half of the JPEG path: option validation, quantiser and Huffman
// tables, JFIF headers and the entropy coder that turns the GPU's coefficient tuple
// into a byte stream identical to the reference's.
//
// This is PRODUCT code (C++17).  It never includes or links anything from oracle/.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pixo_hip.h"

namespace pixo_host {

// ---- quantiser tables: reference src/jpeg/quantize.rs:42-89 ------------------------
struct QuantTables {
    uint8_t lum_zigzag[64]; // DQT payloads (zig-zag order)
    uint8_t chr_zigzag[64];
    float lum[64];          // natural order, exact integers 1..255
    float chr[64];
};
QuantTables make_quant_tables(uint8_t quality);

// Device-side table block (layout documented in jpeg_tile.h: bracketing reciprocals rlo/rhi for
// luminance, q lum, q chr, rlo/rhi chrominance, rlo/rhi chrominance / 4)
constexpr int kDeviceQtFloats = 512;
void fill_device_qt(uint8_t quality, float out[kDeviceQtFloats]);

extern const uint8_t kZigzag[64]; // quantize.rs:18-22

// ---- Huffman tables: reference src/jpeg/huffman.rs ---------------------------------
struct HuffTable {
    uint8_t bits[16];
    uint8_t vals[256];
    int nvals = 0;
    uint16_t code[256];
    uint8_t len[256];
    bool assign_codes(int symbol_limit); // canonical codes, huffman.rs:264-291
};
struct HuffSet {
    HuffTable dc[2]; // [0] luminance, [1] chrominance
    HuffTable ac[2];
    static HuffSet standard();                                       // Annex K, huffman.rs:17-62
    // huffman.rs:167-205 + jpeg/mod.rs:380-390: falls back to standard() like unwrap_or_default
    static HuffSet optimized(const uint64_t dc_counts[2][12], const uint64_t ac_counts[2][256],
                             bool has_chroma);
};

// ---- geometry of the coefficient tuple (jpeg/mod.rs:58-61) --------------------------
struct Geometry {
    bool gray, s420;
    uint32_t units_x, units_y; // MCUs (4:2:0) or 8x8 blocks
    size_t y_blocks, c_blocks, units;
};
Geometry geometry(uint32_t w, uint32_t h, uint8_t color_type, uint8_t subsampling);

// ---- validation in the reference's order (jpeg/mod.rs:333-373) ----------------------
// Returns PIXO_OK or a pixo_status and fills `msg` with the pixo::Error Display text.
int validate(const pixo_jpeg_options &o, bool check_len, size_t data_len, std::string &msg);

// ---- entropy coding -------------------------------------------------------------------
// count_block statistics over the scan (jpeg/mod.rs:684-860), restart resets included.
void symbol_histograms(const int16_t *y, const int16_t *cb, const int16_t *cr,
                       const pixo_jpeg_options &o, uint64_t dc[2][12], uint64_t ac[2][256]);

// Everything of the file before the entropy-coded segment: SOI, APP0, DQT x2, SOF0, DHT x4, [DRI],
// SOS (jpeg/mod.rs:449-648).  Appends to `out`.
void file_headers(std::vector<uint8_t> &out, const pixo_jpeg_options &o, const HuffSet &h);

// The tables as the device entropy stage reads them (jpeg_scan_block.h): 2 x (12 + 256) words,
// (code length << 16) | code, class 0 luminance, class 1 chrominance.
constexpr int kScanTableWords = 2 * (12 + 256);
void pack_scan_tables(const HuffSet &h, uint32_t out[kScanTableWords]);

// Whole file from a coefficient tuple: headers (jpeg/mod.rs:449-648), scan
// (encode_scan :1408-1563 ordering, encode_block huffman.rs:423-481, BitWriterMsb
// bits.rs:195-293 incl. 0xFF stuffing, 1-padding and RSTn), EOI.
void encode_file(const int16_t *y, const int16_t *cb, const int16_t *cr,
                 const pixo_jpeg_options &o, std::vector<uint8_t> &out);

// ---- one image as MCU-row bands (include/pixo_hip.h "band encoder") -------------------
// A band's share of the scan travels as a PIECE: 16 header bytes
//   [0] head_nbits [1] head_bits [2] tail_nbits [3] tail_bits [4..7] 0 [8..15] body_len (u64 LE)
// + body.  With the band starting at bit `bit_offset` of the scan: head = its first (8 - offset % 8) % 8
// bits (they share a byte with the band before), body = its whole bytes, already 0xFF-stuffed, tail =
// the bits left over (they share a byte with the next band); bit values right-aligned.
constexpr size_t kPieceHeader = 16;
void make_piece(std::vector<uint8_t> &piece, int head_n, uint32_t head, int tail_n, uint32_t tail, const uint8_t *body,
                size_t body_len);
// Host twins of the device band encoder, on a band's tuple (`band` = the image's options with the band's
// height; restart intervals do not apply to bands); prev_dc = last DC of Y, Cb, Cr above the band.
void band_histograms(const int16_t *y, const int16_t *cb, const int16_t *cr, const pixo_jpeg_options &band,
                     const int16_t prev_dc[3], uint64_t dc[2][12], uint64_t ac[2][256]);
uint64_t band_bits(const int16_t *y, const int16_t *cb, const int16_t *cr, const pixo_jpeg_options &band, const HuffSet &h,
                   const int16_t prev_dc[3]);
void band_piece(const int16_t *y, const int16_t *cb, const int16_t *cr, const pixo_jpeg_options &band, const HuffSet &h,
                const int16_t prev_dc[3], uint64_t bit_offset, std::vector<uint8_t> &piece);
// Where everything goes in the finished file, from the 16-byte piece headers alone: the JFIF headers, every
// band's body (so that it can be copied straight to its final place, e.g. by its own GPU), and the bytes
// in between — bytes two bands share (merged, stuffed), the final 1-padding, EOI — as (position, value).
struct SpliceLayout {
    std::vector<uint8_t> head;
    std::vector<size_t> body_off, body_len;
    std::vector<std::pair<size_t, uint8_t>> fixups;
    size_t file_len = 0;
};
int splice_layout(const pixo_jpeg_options &o, const HuffSet &h, const uint8_t *piece_headers, uint32_t parts, SpliceLayout &l,
                  std::string &msg);
void splice_finish(const SpliceLayout &l, uint8_t *file); // headers + fixups (the bodies are the caller's)
// headers + the pieces merged bit-exactly (shared bytes OR-ed and stuffed, final 1-padding) + EOI
int splice_file(const pixo_jpeg_options &o, const HuffSet &h, const uint8_t *const *pieces, const size_t *lens, uint32_t parts,
                std::vector<uint8_t> &out, std::string &msg);

// Progressive file from a coefficient tuple and the tables to use: SOF2 headers, the seven scans of
// simple_progressive_script (src/jpeg/progressive.rs:98-110) coded like jpeg/mod.rs:1248-1365, EOI.
void encode_progressive_file(const int16_t *y, const int16_t *cb, const int16_t *cr, const pixo_jpeg_options &o,
                             const HuffSet &h, std::vector<uint8_t> &out);

} // namespace pixo_host
