// jpeg_host.cpp — see jpeg_host.hpp.  Product code; independent of oracle/.
#include "jpeg_host.hpp"

#include <cmath>
#include <algorithm>
#include <cstring>
#include <functional>
#include <queue>
#include <thread>
#include <utility>

namespace pixo_host {

// ======================================================================================
// quantiser tables (quantize.rs:4-89)
// ======================================================================================
namespace {
constexpr uint8_t kBaseLum[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
constexpr uint8_t kBaseChr[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

inline uint32_t scale_entry(uint32_t base, uint32_t scale)
{
    return std::clamp<uint32_t>((base * scale + 50) / 100, 1, 255);
}
} // namespace

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                             12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                             58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

QuantTables make_quant_tables(uint8_t quality)
{
    const uint32_t q = std::clamp<uint32_t>(quality, 1, 100);
    const uint32_t scale = q < 50 ? 5000 / q : 200 - 2 * q; // libjpeg formula, quantize.rs:46-50
    QuantTables t;
    for (int i = 0; i < 64; ++i) {
        t.lum[i] = static_cast<float>(scale_entry(kBaseLum[i], scale));
        t.chr[i] = static_cast<float>(scale_entry(kBaseChr[i], scale));
    }
    for (int i = 0; i < 64; ++i) {
        t.lum_zigzag[i] = static_cast<uint8_t>(t.lum[kZigzag[i]]);
        t.chr_zigzag[i] = static_cast<uint8_t>(t.chr[kZigzag[i]]);
    }
    return t;
}

namespace {
// Directed roundings of (1/q)(1 -+ (2^-24 + 2^-30)): the largest f32 not above / the smallest f32 not below
// (jpeg_tile.h quant_row8 proves the bracket from exactly these two properties).
float bracket_lo(float q)
{
    const double want = (1.0 / q) * (1.0 - 0x1p-24 - 0x1p-30);
    float f = static_cast<float>(want);
    while (static_cast<double>(f) > want) f = std::nextafterf(f, 0.0f);
    return f;
}
float bracket_hi(float q)
{
    const double want = (1.0 / q) * (1.0 + 0x1p-24 + 0x1p-30);
    float f = static_cast<float>(want);
    while (static_cast<double>(f) < want) f = std::nextafterf(f, 2.0f);
    return f;
}
} // namespace

void fill_device_qt(uint8_t quality, float out[kDeviceQtFloats])
{
    const QuantTables t = make_quant_tables(quality);
    for (int i = 0; i < 64; ++i) {
        out[2 * i] = bracket_lo(t.lum[i]); // (rlo, rhi) side by side: one operand pair of the kernel's packed multiply-add
        out[2 * i + 1] = bracket_hi(t.lum[i]);
        out[128 + i] = t.lum[i];
        out[192 + i] = t.chr[i];
        out[256 + 2 * i] = bracket_lo(t.chr[i]);
        out[256 + 2 * i + 1] = bracket_hi(t.chr[i]);
        // 4:2:0 chroma is transformed at 4x scale and NEGATED (the kernel's planes hold 255 - Cb, jpeg_tile.h
        // color_row4_dot): exact factor -1/4; the two products still bracket the reference quotient
        out[384 + 2 * i] = out[256 + 2 * i] * -0.25f;
        out[384 + 2 * i + 1] = out[256 + 2 * i + 1] * -0.25f;
    }
}

// ======================================================================================
// Huffman tables
// ======================================================================================
namespace {
// JPEG Annex K.3 typical tables (the reference ships the same data, huffman.rs:17-62)
constexpr uint8_t kDcLumBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
constexpr uint8_t kDcChrBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
constexpr uint8_t kAcLumBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125};
constexpr uint8_t kAcChrBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119};
constexpr uint8_t kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
constexpr uint8_t kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

void load_spec(HuffTable &t, const uint8_t bits[16], const uint8_t *vals, int n)
{
    std::memcpy(t.bits, bits, 16);
    std::memcpy(t.vals, vals, static_cast<size_t>(n));
    t.nvals = n;
}

// Code lengths from symbol counts, as the reference builds them (huffman.rs:317-391):
// leaves in ascending symbol order, min-heap on (frequency, node id), merged node gets
// the next id, and the emitted length is tree depth + 1.  Returns false for an empty
// histogram or when a length would exceed 16 (the reference returns None).
bool code_lengths(const uint64_t *counts, int n, uint8_t *lengths)
{
    struct Node { int left, right, symbol; };
    std::vector<Node> nodes;
    using Key = std::pair<uint64_t, int>;
    std::priority_queue<Key, std::vector<Key>, std::greater<Key>> heap;
    std::fill(lengths, lengths + n, uint8_t{0});
    for (int s = 0; s < n; ++s)
        if (counts[s] != 0) {
            heap.emplace(counts[s], static_cast<int>(nodes.size()));
            nodes.push_back({-1, -1, s});
        }
    if (heap.empty()) return false;
    if (heap.size() == 1) {
        lengths[nodes[heap.top().second].symbol] = 1;
        return true;
    }
    while (heap.size() > 1) {
        const Key a = heap.top(); heap.pop();
        const Key b = heap.top(); heap.pop();
        heap.emplace(a.first + b.first, static_cast<int>(nodes.size()));
        nodes.push_back({a.second, b.second, -1});
    }
    std::vector<std::pair<int, int>> todo{{heap.top().second, 0}};
    while (!todo.empty()) {
        const auto [id, depth] = todo.back();
        todo.pop_back();
        const Node &nd = nodes[static_cast<size_t>(id)];
        if (nd.symbol >= 0) {
            if (depth + 1 > 16) return false;
            lengths[nd.symbol] = static_cast<uint8_t>(depth + 1);
        } else {
            todo.emplace_back(nd.left, depth + 1);
            todo.emplace_back(nd.right, depth + 1);
        }
    }
    return true;
}

// BITS/VALS from counts (huffman.rs:294-315): vals ordered by (length, symbol).
bool spec_from_counts(const uint64_t *counts, int n, HuffTable &t)
{
    uint8_t len[256];
    if (!code_lengths(counts, n, len)) return false;
    std::memset(t.bits, 0, 16);
    t.nvals = 0;
    for (int l = 1; l <= 16; ++l)
        for (int s = 0; s < n; ++s)
            if (len[s] == l) {
                ++t.bits[l - 1];
                t.vals[t.nvals++] = static_cast<uint8_t>(s);
            }
    return true;
}
} // namespace

bool HuffTable::assign_codes(int symbol_limit)
{
    std::memset(code, 0, sizeof code);
    std::memset(len, 0, sizeof len);
    uint32_t next = 0;
    int vi = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int k = 0; k < bits[l - 1]; ++k, ++vi, ++next) {
            if (vi >= nvals || vals[vi] >= symbol_limit) return false;
            code[vals[vi]] = static_cast<uint16_t>(next);
            len[vals[vi]] = static_cast<uint8_t>(l);
        }
        next <<= 1;
    }
    return true;
}

HuffSet HuffSet::standard()
{
    static const uint8_t dc_vals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
    HuffSet h;
    load_spec(h.dc[0], kDcLumBits, dc_vals, 12);
    load_spec(h.dc[1], kDcChrBits, dc_vals, 12);
    load_spec(h.ac[0], kAcLumBits, kAcLumVals, 162);
    load_spec(h.ac[1], kAcChrBits, kAcChrVals, 162);
    h.dc[0].assign_codes(12);
    h.dc[1].assign_codes(12);
    h.ac[0].assign_codes(256);
    h.ac[1].assign_codes(256);
    return h;
}

HuffSet HuffSet::optimized(const uint64_t dc_counts[2][12], const uint64_t ac_counts[2][256],
                           bool has_chroma)
{
    HuffSet h = standard();
    // luminance tables are mandatory: any failure voids the whole optimised set
    if (!spec_from_counts(dc_counts[0], 12, h.dc[0]) || !spec_from_counts(ac_counts[0], 256, h.ac[0]))
        return standard();
    if (has_chroma) { // chroma tables fall back individually (huffman.rs:176-185)
        HuffTable t;
        if (spec_from_counts(dc_counts[1], 12, t)) { load_spec(h.dc[1], t.bits, t.vals, t.nvals); }
        if (spec_from_counts(ac_counts[1], 256, t)) { load_spec(h.ac[1], t.bits, t.vals, t.nvals); }
    }
    if (!h.dc[0].assign_codes(12) || !h.dc[1].assign_codes(12) || !h.ac[0].assign_codes(256) ||
        !h.ac[1].assign_codes(256))
        return standard();
    return h;
}

// ======================================================================================
// geometry + validation
// ======================================================================================
Geometry geometry(uint32_t w, uint32_t h, uint8_t color_type, uint8_t subsampling)
{
    Geometry g{};
    g.gray = color_type == PIXO_GRAY;
    g.s420 = !g.gray && subsampling == PIXO_S420;
    const uint32_t unit = g.s420 ? 16 : 8;
    g.units_x = (w + unit - 1) / unit;
    g.units_y = (h + unit - 1) / unit;
    g.units = static_cast<size_t>(g.units_x) * g.units_y;
    g.y_blocks = g.s420 ? 4 * g.units : g.units;
    g.c_blocks = g.gray ? 0 : g.units;
    return g;
}

int validate(const pixo_jpeg_options &o, bool check_len, size_t data_len, std::string &msg)
{
    // order and wording: src/jpeg/mod.rs:333-373, src/error.rs:50-91
    if (o.quality == 0 || o.quality > 100) {
        msg = "Invalid quality " + std::to_string(o.quality) + ": must be 1-100";
        return PIXO_ERR_INVALID_QUALITY;
    }
    if (o.has_restart_interval && o.restart_interval == 0) {
        msg = "Invalid restart interval 0: must be 1-65535 (or None to disable)";
        return PIXO_ERR_INVALID_RESTART_INTERVAL;
    }
    if (o.width == 0 || o.height == 0) {
        msg = "Invalid image dimensions: " + std::to_string(o.width) + "x" + std::to_string(o.height);
        return PIXO_ERR_INVALID_DIMENSIONS;
    }
    if (o.width > 65535 || o.height > 65535) {
        msg = "Image " + std::to_string(o.width) + "x" + std::to_string(o.height) +
              " exceeds maximum dimension 65535";
        return PIXO_ERR_IMAGE_TOO_LARGE;
    }
    if (o.color_type != PIXO_RGB && o.color_type != PIXO_GRAY) {
        msg = "Unsupported color type for this format";
        return PIXO_ERR_UNSUPPORTED_COLOR_TYPE;
    }
    if (check_len) {
        const size_t want = static_cast<size_t>(o.width) * o.height * (o.color_type == PIXO_RGB ? 3 : 1);
        if (data_len != want) {
            msg = "Invalid pixel data length: expected " + std::to_string(want) + " bytes, got " +
                  std::to_string(data_len);
            return PIXO_ERR_INVALID_DATA_LENGTH;
        }
    }
    return PIXO_OK;
}

// ======================================================================================
// scan walking: block order and DC prediction of encode_scan (jpeg/mod.rs:1448-1557)
// ======================================================================================
namespace {

inline int magnitude_bits(int v)
{ // huffman.rs:394-401 `category`
    const unsigned a = static_cast<unsigned>(v < 0 ? -v : v);
    return a == 0 ? 0 : 32 - __builtin_clz(a);
}

// MSB-first bit sink with JPEG byte stuffing (bits.rs:195-293), 64-bit accumulator.
class BitSink {
  public:
    explicit BitSink(std::vector<uint8_t> &out) : out_(out) {}
    inline void put(uint32_t value, int nbits)
    { // nbits <= 32, pending_ < 32 on entry
        acc_ = (acc_ << nbits) | value;
        pending_ += nbits;
        if (pending_ >= 32) drain();
    }
    void align_with_ones()
    { // bits.rs:261-272 flush(): pad the partial byte with 1s
        drain();
        if (pending_ > 0) {
            const int pad = 8 - pending_;
            acc_ = (acc_ << pad) | ((1u << pad) - 1u);
            pending_ = 8;
            drain();
        }
    }
    void raw(uint8_t a, uint8_t b) { out_.push_back(a); out_.push_back(b); } // byte aligned only
  private:
    void drain()
    {
        while (pending_ >= 8) {
            const uint8_t byte = static_cast<uint8_t>(acc_ >> (pending_ - 8));
            out_.push_back(byte);
            if (byte == 0xFF) out_.push_back(0x00);
            pending_ -= 8;
        }
        acc_ &= (uint64_t{1} << pending_) - 1;
    }
    std::vector<uint8_t> &out_;
    uint64_t acc_ = 0;
    int pending_ = 0;
};

struct Emit {
    BitSink &sink;
    const HuffSet &h;
    inline int16_t block(const int16_t *blk, int16_t prev_dc, int cls)
    { // encode_block, huffman.rs:423-481
        const HuffTable &dc = h.dc[cls], &ac = h.ac[cls];
        const int16_t d0 = blk[0];
        const int diff = static_cast<int16_t>(d0 - prev_dc);
        const int dcat = magnitude_bits(diff);
        sink.put(dc.code[dcat], dc.len[dcat]);
        if (dcat) sink.put(static_cast<uint32_t>(diff < 0 ? diff - 1 : diff) & ((1u << dcat) - 1u), dcat);
        int run = 0;
        for (int k = 1; k < 64; ++k) {
            const int v = blk[kZigzag[k]];
            if (v == 0) { ++run; continue; }
            for (; run >= 16; run -= 16) sink.put(ac.code[0xF0], ac.len[0xF0]);
            const int cat = magnitude_bits(v);
            const int rs = (run << 4) | cat;
            const uint32_t vb = static_cast<uint32_t>(v < 0 ? v - 1 : v) & ((1u << cat) - 1u);
            sink.put((static_cast<uint32_t>(ac.code[rs]) << cat) | vb, ac.len[rs] + cat);
            run = 0;
        }
        if (run > 0) sink.put(ac.code[0], ac.len[0]);
        return d0;
    }
    void restart(uint8_t idx) { sink.align_with_ones(); sink.raw(0xFF, static_cast<uint8_t>(0xD0 + (idx & 7))); }
};

struct Count {
    uint64_t (*dc)[12];
    uint64_t (*ac)[256];
    inline int16_t block(const int16_t *blk, int16_t prev_dc, int cls)
    { // count_block, jpeg/mod.rs:826-860
        const int16_t d0 = blk[0];
        ++dc[cls][magnitude_bits(static_cast<int16_t>(d0 - prev_dc))];
        int run = 0;
        for (int k = 1; k < 64; ++k) {
            const int v = blk[kZigzag[k]];
            if (v == 0) { ++run; continue; }
            for (; run >= 16; run -= 16) ++ac[cls][0xF0];
            ++ac[cls][((run << 4) | magnitude_bits(v)) & 0xFF];
            run = 0;
        }
        if (run > 0) ++ac[cls][0];
        return d0;
    }
    void restart(uint8_t) {}
};

template <class Visitor>
void walk_scan(Visitor &vis, const int16_t *y, const int16_t *cb, const int16_t *cr,
               const pixo_jpeg_options &o, const int16_t *seed_dc = nullptr)
{
    const Geometry g = geometry(o.width, o.height, o.color_type, o.subsampling);
    // (seed_dc: the predictors a band of a larger image starts from — the DCs of the band above)
    int16_t py = seed_dc ? seed_dc[0] : 0, pcb = seed_dc ? seed_dc[1] : 0, pcr = seed_dc ? seed_dc[2] : 0;
    uint8_t rst = 0;
    const uint32_t total = static_cast<uint32_t>(g.units);
    const uint32_t interval = o.has_restart_interval ? o.restart_interval : 0;
    for (uint32_t m = 0; m < total; ++m) {
        if (g.gray) {
            py = vis.block(y + size_t{m} * 64, py, 0);
        } else if (!g.s420) {
            py = vis.block(y + size_t{m} * 64, py, 0);
            pcb = vis.block(cb + size_t{m} * 64, pcb, 1);
            pcr = vis.block(cr + size_t{m} * 64, pcr, 1);
        } else {
            const int16_t *yb = y + size_t{m} * 256;
            for (int k = 0; k < 4; ++k) py = vis.block(yb + k * 64, py, 0);
            pcb = vis.block(cb + size_t{m} * 64, pcb, 1);
            pcr = vis.block(cr + size_t{m} * 64, pcr, 1);
        }
        // restart only when more MCUs follow (jpeg/mod.rs:1431-1445)
        if (interval && (m + 1) % interval == 0 && m + 1 < total) {
            vis.restart(rst);
            rst = static_cast<uint8_t>((rst + 1) & 7);
            py = pcb = pcr = 0;
        }
    }
}

void be16(std::vector<uint8_t> &v, unsigned x)
{
    v.push_back(static_cast<uint8_t>(x >> 8));
    v.push_back(static_cast<uint8_t>(x));
}

void dht_segment(std::vector<uint8_t> &v, uint8_t id, const HuffTable &t)
{
    be16(v, 0xFFC4);
    be16(v, static_cast<unsigned>(2 + 1 + 16 + t.nvals));
    v.push_back(id);
    v.insert(v.end(), t.bits, t.bits + 16);
    v.insert(v.end(), t.vals, t.vals + t.nvals);
}

// SOI, APP0 (JFIF 1.01, no units, 1x1), DQT x2, SOF0, DHT x4, [DRI], SOS — jpeg/mod.rs:449-648
void write_headers(std::vector<uint8_t> &v, const pixo_jpeg_options &o, const QuantTables &qt,
                   const HuffSet &h)
{
    be16(v, 0xFFD8);
    static const uint8_t jfif[] = {0xFF, 0xE0, 0x00, 0x10, 'J', 'F', 'I', 'F', 0x00, 0x01, 0x01, 0x00, 0x00, 0x01, 0x00, 0x01, 0x00, 0x00};
    v.insert(v.end(), jfif, jfif + sizeof jfif);
    for (int id = 0; id < 2; ++id) {
        be16(v, 0xFFDB);
        be16(v, 67);
        v.push_back(static_cast<uint8_t>(id));
        const uint8_t *tab = id == 0 ? qt.lum_zigzag : qt.chr_zigzag;
        v.insert(v.end(), tab, tab + 64);
    }
    const bool gray = o.color_type == PIXO_GRAY;
    const unsigned ncomp = gray ? 1 : 3;
    be16(v, o.progressive ? 0xFFC2 : 0xFFC0); // SOF2 for progressive scans (jpeg/mod.rs:397-400, :508-516)
    be16(v, 8 + 3 * ncomp);
    v.push_back(8);
    be16(v, o.height & 0xFFFF);
    be16(v, o.width & 0xFFFF);
    v.push_back(static_cast<uint8_t>(ncomp));
    v.push_back(1);
    v.push_back(!gray && o.subsampling == PIXO_S420 ? 0x22 : 0x11);
    v.push_back(0);
    if (!gray) {
        for (uint8_t id = 2; id <= 3; ++id) { v.push_back(id); v.push_back(0x11); v.push_back(1); }
    }
    dht_segment(v, 0x00, h.dc[0]);
    dht_segment(v, 0x01, h.dc[1]);
    dht_segment(v, 0x10, h.ac[0]);
    dht_segment(v, 0x11, h.ac[1]);
    if (o.has_restart_interval) { be16(v, 0xFFDD); be16(v, 4); be16(v, o.restart_interval); }
    if (o.progressive) return; // every progressive scan writes its own SOS
    be16(v, 0xFFDA);
    be16(v, 6 + 2 * ncomp);
    v.push_back(static_cast<uint8_t>(ncomp));
    v.push_back(1); v.push_back(0x00);
    if (!gray) { v.push_back(2); v.push_back(0x11); v.push_back(3); v.push_back(0x11); }
    v.push_back(0); v.push_back(63); v.push_back(0);
}
// ---- progressive scans (jpeg/mod.rs:872-927, :1248-1365; progressive.rs) -------------------------
// A symbol the table does not contain is coded as (0, 4 bits): progressive.rs:363-381 falls back to
// that instead of failing, and the end-of-band run symbols 0x10..0xE0 are never in these tables.
struct Code { uint32_t code; int len; };
inline Code code_of(const HuffTable &t, int symbol)
{
    return t.len[symbol & 0xFF] ? Code{t.code[symbol & 0xFF], t.len[symbol & 0xFF]} : Code{0, 4};
}

void flush_band_run(BitSink &sink, unsigned &run, const HuffTable &ac)
{ // progressive.rs:313-345: symbol = floor(log2 run) << 4, then the low bits of the run
    if (run == 0) return;
    const int nbits = 31 - __builtin_clz(run);
    const Code c = code_of(ac, nbits << 4);
    sink.put(c.code, c.len);
    if (nbits > 0) sink.put(run - (1u << nbits), nbits);
    run = 0;
}

// One block of a first AC scan over zig-zag positions [ss, se] (progressive.rs:141-210, al = 0)
void ac_band_block(BitSink &sink, const int16_t *blk, int ss, int se, unsigned &run, const HuffTable &ac)
{
    int last = se;
    while (last > ss && blk[kZigzag[last]] == 0) --last;
    if (last == ss && blk[kZigzag[ss]] == 0) { // nothing in the band: lengthen the end-of-band run
        if (++run == 0x7FFF) flush_band_run(sink, run, ac);
        return;
    }
    flush_band_run(sink, run, ac);
    int zeros = 0;
    for (int k = ss; k <= last; ++k) {
        const int v = blk[kZigzag[k]];
        if (v == 0) { ++zeros; continue; }
        for (; zeros >= 16; zeros -= 16) { const Code z = code_of(ac, 0xF0); sink.put(z.code, z.len); }
        const int cat = magnitude_bits(v);
        const Code c = code_of(ac, (zeros << 4) | cat);
        sink.put(c.code, c.len);
        sink.put(static_cast<uint32_t>(v < 0 ? v - 1 : v) & ((1u << cat) - 1u), cat);
        zeros = 0;
    }
    if (last < se) run = 1;
}

// simple_progressive_script (progressive.rs:98-110): DC of Y, Cb, Cr, then Y AC 1-10, Y AC 11-63, Cb AC,
// Cr AC; one component per scan, its blocks in STORAGE order (jpeg/mod.rs:1286, :1350), one bit
// stream per scan (flushed with 1-padding).  Gray images still get the chroma SOS headers (:888-925).
void progressive_scans(std::vector<uint8_t> &out, const int16_t *y, const int16_t *cb, const int16_t *cr,
                       const Geometry &g, const HuffSet &h)
{
    static const struct { int comp, ss, se; } script[7] = {{0, 0, 0}, {1, 0, 0}, {2, 0, 0}, {0, 1, 10}, {0, 11, 63}, {1, 1, 63}, {2, 1, 63}};
    // The scans share nothing (each has its own predictor, run counter and bit stream): one thread per
    // scan, results concatenated in script order.
    std::vector<uint8_t> part[7];
    auto run_scan = [&](int i) {
        const auto &sc = script[i];
        std::vector<uint8_t> &o = part[i];
        be16(o, 0xFFDA); be16(o, 8); o.push_back(1); // write_sos_progressive, jpeg/mod.rs:650-682
        o.push_back(static_cast<uint8_t>(sc.comp + 1));
        o.push_back(sc.comp == 0 ? 0x00 : 0x11);
        o.push_back(static_cast<uint8_t>(sc.ss)); o.push_back(static_cast<uint8_t>(sc.se)); o.push_back(0);
        const int16_t *coef = sc.comp == 0 ? y : (sc.comp == 1 ? cb : cr);
        const size_t n = sc.comp == 0 ? g.y_blocks : g.c_blocks;
        const int cls = sc.comp == 0 ? 0 : 1;
        if (n == 0) return;
        o.reserve(n * (sc.se == 0 ? 2 : 24) + 64);
        BitSink sink(o);
        if (sc.se == 0) { // DC scan: differences in storage order, predictor starts at 0 per scan
            int16_t prev = 0;
            for (size_t b = 0; b < n; ++b) {
                const int16_t dc = coef[b * 64];
                const int diff = static_cast<int16_t>(dc - prev);
                const int cat = magnitude_bits(diff);
                const Code c = code_of(h.dc[cls], cat);
                sink.put(c.code, c.len);
                if (cat) sink.put(static_cast<uint32_t>(diff < 0 ? diff - 1 : diff) & ((1u << cat) - 1u), cat);
                prev = dc;
            }
        } else {
            unsigned run = 0;
            for (size_t b = 0; b < n; ++b) ac_band_block(sink, coef + b * 64, sc.ss, sc.se, run, h.ac[cls]);
            flush_band_run(sink, run, h.ac[cls]);
        }
        sink.align_with_ones();
    };
    if (g.y_blocks + 2 * g.c_blocks >= 4096) {
        std::thread workers[7];
        for (int i = 0; i < 7; ++i) workers[i] = std::thread(run_scan, i);
        for (auto &t : workers) t.join();
    } else {
        for (int i = 0; i < 7; ++i) run_scan(i);
    }
    for (const auto &p : part) out.insert(out.end(), p.begin(), p.end());
}
} // namespace

void encode_progressive_file(const int16_t *y, const int16_t *cb, const int16_t *cr, const pixo_jpeg_options &o,
                             const HuffSet &h, std::vector<uint8_t> &out)
{
    const Geometry g = geometry(o.width, o.height, o.color_type, o.subsampling);
    out.clear();
    write_headers(out, o, make_quant_tables(o.quality), h);
    progressive_scans(out, y, cb, cr, g, h);
    be16(out, 0xFFD9);
}

void symbol_histograms(const int16_t *y, const int16_t *cb, const int16_t *cr,
                       const pixo_jpeg_options &o, uint64_t dc[2][12], uint64_t ac[2][256])
{
    std::memset(dc, 0, sizeof(uint64_t) * 24);
    std::memset(ac, 0, sizeof(uint64_t) * 512);
    Count c{dc, ac};
    walk_scan(c, y, cb, cr, o);
}

void file_headers(std::vector<uint8_t> &out, const pixo_jpeg_options &o, const HuffSet &h)
{
    write_headers(out, o, make_quant_tables(o.quality), h);
}

void pack_scan_tables(const HuffSet &h, uint32_t out[kScanTableWords])
{
    for (int cls = 0; cls < 2; ++cls) {
        uint32_t *t = out + cls * (12 + 256);
        for (int i = 0; i < 12; ++i) t[i] = (static_cast<uint32_t>(h.dc[cls].len[i]) << 16) | h.dc[cls].code[i];
        for (int i = 0; i < 256; ++i) t[12 + i] = (static_cast<uint32_t>(h.ac[cls].len[i]) << 16) | h.ac[cls].code[i];
    }
}

// ======================================================================================
// One image as MCU-row bands: every band is entropy-coded on its own — with the DC
// predictors of the band above and at its bit offset in the scan — and the pieces are spliced.
// ======================================================================================
namespace {
// bit sinks for walk_scan's Emit: count only / collect the raw (unstuffed) bits
struct BitCounter {
    uint64_t bits = 0;
    inline void put(uint32_t, int nbits) { bits += static_cast<uint64_t>(nbits); }
    void align_with_ones() {}
    void raw(uint8_t, uint8_t) {}
};
struct RawBits {
    std::vector<uint8_t> bytes; // MSB first, last byte zero padded
    uint64_t bits = 0;
    inline void put(uint32_t value, int nbits)
    {
        for (int i = nbits - 1; i >= 0; --i) {
            if ((bits & 7) == 0) bytes.push_back(0);
            if ((value >> i) & 1u) bytes.back() |= static_cast<uint8_t>(0x80u >> (bits & 7));
            ++bits;
        }
    }
    void align_with_ones() {}
    void raw(uint8_t, uint8_t) {}
    uint32_t get(uint64_t first, int n) const // n <= 8 bits starting at bit `first`
    {
        uint32_t v = 0;
        for (int i = 0; i < n; ++i) {
            const uint64_t b = first + i;
            v = (v << 1) | ((bytes[b >> 3] >> (7 - (b & 7))) & 1u);
        }
        return v;
    }
};
template <class Sink> struct EmitTo {
    Sink &sink;
    const HuffSet &h;
    inline int16_t block(const int16_t *blk, int16_t prev_dc, int cls)
    { // encode_block, huffman.rs:423-481 (same walk as Emit)
        const HuffTable &dc = h.dc[cls], &ac = h.ac[cls];
        const int16_t d0 = blk[0];
        const int diff = static_cast<int16_t>(d0 - prev_dc);
        const int dcat = magnitude_bits(diff);
        sink.put(dc.code[dcat], dc.len[dcat]);
        if (dcat) sink.put(static_cast<uint32_t>(diff < 0 ? diff - 1 : diff) & ((1u << dcat) - 1u), dcat);
        int run = 0;
        for (int k = 1; k < 64; ++k) {
            const int v = blk[kZigzag[k]];
            if (v == 0) { ++run; continue; }
            for (; run >= 16; run -= 16) sink.put(ac.code[0xF0], ac.len[0xF0]);
            const int cat = magnitude_bits(v);
            const int rs = (run << 4) | cat;
            const uint32_t vb = static_cast<uint32_t>(v < 0 ? v - 1 : v) & ((1u << cat) - 1u);
            sink.put((static_cast<uint32_t>(ac.code[rs]) << cat) | vb, ac.len[rs] + cat);
            run = 0;
        }
        if (run > 0) sink.put(ac.code[0], ac.len[0]);
        return d0;
    }
    void restart(uint8_t) {}
};
pixo_jpeg_options without_restart(pixo_jpeg_options o)
{
    o.has_restart_interval = 0; o.restart_interval = 0;
    return o;
}
} // namespace

void make_piece(std::vector<uint8_t> &piece, int head_n, uint32_t head, int tail_n, uint32_t tail, const uint8_t *body,
                size_t body_len)
{
    piece.assign(kPieceHeader + body_len, 0);
    piece[0] = static_cast<uint8_t>(head_n); piece[1] = static_cast<uint8_t>(head);
    piece[2] = static_cast<uint8_t>(tail_n); piece[3] = static_cast<uint8_t>(tail);
    for (int i = 0; i < 8; ++i) piece[8 + i] = static_cast<uint8_t>(static_cast<uint64_t>(body_len) >> (8 * i));
    if (body_len) std::memcpy(piece.data() + kPieceHeader, body, body_len);
}

void band_histograms(const int16_t *y, const int16_t *cb, const int16_t *cr, const pixo_jpeg_options &band,
                     const int16_t prev_dc[3], uint64_t dc[2][12], uint64_t ac[2][256])
{
    std::memset(dc, 0, sizeof(uint64_t) * 24);
    std::memset(ac, 0, sizeof(uint64_t) * 512);
    Count c{dc, ac};
    walk_scan(c, y, cb, cr, without_restart(band), prev_dc);
}

uint64_t band_bits(const int16_t *y, const int16_t *cb, const int16_t *cr, const pixo_jpeg_options &band, const HuffSet &h,
                   const int16_t prev_dc[3])
{
    BitCounter sink;
    EmitTo<BitCounter> e{sink, h};
    walk_scan(e, y, cb, cr, without_restart(band), prev_dc);
    return sink.bits;
}

void band_piece(const int16_t *y, const int16_t *cb, const int16_t *cr, const pixo_jpeg_options &band, const HuffSet &h,
                const int16_t prev_dc[3], uint64_t bit_offset, std::vector<uint8_t> &piece)
{
    RawBits sink;
    EmitTo<RawBits> e{sink, h};
    walk_scan(e, y, cb, cr, without_restart(band), prev_dc);
    const uint64_t bits = sink.bits;
    const int want = static_cast<int>((8 - (bit_offset & 7)) & 7);
    const int head_n = static_cast<int>(bits < static_cast<uint64_t>(want) ? bits : want);
    const uint64_t rest = bits - head_n, nfull = rest / 8;
    const int tail_n = static_cast<int>(rest % 8);
    std::vector<uint8_t> body;
    body.reserve(nfull + nfull / 64 + 8);
    for (uint64_t i = 0; i < nfull; ++i) {
        const uint8_t b = static_cast<uint8_t>(sink.get(head_n + 8 * i, 8));
        body.push_back(b);
        if (b == 0xFF) body.push_back(0x00); // bits.rs:245-253
    }
    make_piece(piece, head_n, head_n ? sink.get(0, head_n) : 0, tail_n, tail_n ? sink.get(head_n + 8 * nfull, tail_n) : 0,
               body.data(), body.size());
}

int splice_layout(const pixo_jpeg_options &o, const HuffSet &h, const uint8_t *piece_headers, uint32_t parts, SpliceLayout &l,
                  std::string &msg)
{
    l.head.clear();
    write_headers(l.head, o, make_quant_tables(o.quality), h);
    l.body_off.assign(parts, 0);
    l.body_len.assign(parts, 0);
    l.fixups.clear();
    size_t pos = l.head.size();
    uint32_t acc = 0;
    int nacc = 0; // bits of the byte two neighbouring bands share
    auto emit = [&](uint8_t b) {
        l.fixups.emplace_back(pos++, b);
        if (b == 0xFF) l.fixups.emplace_back(pos++, static_cast<uint8_t>(0x00)); // bits.rs:245-253
    };
    for (uint32_t k = 0; k < parts; ++k) {
        const uint8_t *p = piece_headers + size_t{k} * kPieceHeader;
        const int head_n = p[0], tail_n = p[2];
        uint64_t body_len = 0;
        for (int i = 0; i < 8; ++i) body_len |= static_cast<uint64_t>(p[8 + i]) << (8 * i);
        if (head_n > 7 || tail_n > 7 || nacc + head_n > 8 || ((body_len || tail_n) && nacc + head_n != 8 && nacc + head_n != 0)) {
            msg = "Compression error: band piece " + std::to_string(k) + " does not start at the bit offset the bands before it end at";
            return PIXO_ERR_COMPRESSION;
        }
        if (head_n) {
            acc = (acc << head_n) | (p[1] & ((1u << head_n) - 1u));
            nacc += head_n;
            if (nacc == 8) { emit(static_cast<uint8_t>(acc)); acc = 0; nacc = 0; }
        }
        l.body_off[k] = pos;
        l.body_len[k] = static_cast<size_t>(body_len);
        pos += static_cast<size_t>(body_len);
        if (tail_n) { acc = p[3] & ((1u << tail_n) - 1u); nacc = tail_n; }
    }
    if (nacc) emit(static_cast<uint8_t>((acc << (8 - nacc)) | ((1u << (8 - nacc)) - 1u))); // BitWriterMsb::flush, bits.rs:261-272
    l.fixups.emplace_back(pos++, static_cast<uint8_t>(0xFF)); // EOI
    l.fixups.emplace_back(pos++, static_cast<uint8_t>(0xD9));
    l.file_len = pos;
    return PIXO_OK;
}

void splice_finish(const SpliceLayout &l, uint8_t *file)
{
    std::memcpy(file, l.head.data(), l.head.size());
    for (const auto &f : l.fixups) file[f.first] = f.second;
}

int splice_file(const pixo_jpeg_options &o, const HuffSet &h, const uint8_t *const *pieces, const size_t *lens, uint32_t parts,
                std::vector<uint8_t> &out, std::string &msg)
{
    std::vector<uint8_t> headers(size_t{parts} * kPieceHeader);
    for (uint32_t k = 0; k < parts; ++k) {
        if (!pieces[k] || lens[k] < kPieceHeader) { msg = "Compression error: band piece " + std::to_string(k) + " is malformed"; return PIXO_ERR_COMPRESSION; }
        std::memcpy(headers.data() + size_t{k} * kPieceHeader, pieces[k], kPieceHeader);
    }
    SpliceLayout l;
    int rc = splice_layout(o, h, headers.data(), parts, l, msg);
    if (rc) return rc;
    for (uint32_t k = 0; k < parts; ++k)
        if (l.body_len[k] != lens[k] - kPieceHeader) { msg = "Compression error: band piece " + std::to_string(k) + " is malformed"; return PIXO_ERR_COMPRESSION; }
    out.assign(l.file_len, 0);
    for (uint32_t k = 0; k < parts; ++k)
        if (l.body_len[k]) std::memcpy(out.data() + l.body_off[k], pieces[k] + kPieceHeader, l.body_len[k]);
    splice_finish(l, out.data());
    return PIXO_OK;
}

void encode_file(const int16_t *y, const int16_t *cb, const int16_t *cr,
                 const pixo_jpeg_options &o, std::vector<uint8_t> &out)
{
    const QuantTables qt = make_quant_tables(o.quality);
    HuffSet h;
    if (o.optimize_huffman) {
        uint64_t dc[2][12], ac[2][256];
        symbol_histograms(y, cb, cr, o, dc, ac);
        h = HuffSet::optimized(dc, ac, o.color_type != PIXO_GRAY);
    } else {
        h = HuffSet::standard();
    }
    if (o.progressive) { // (statistics for optimised tables: those of a baseline scan over this tuple)
        encode_progressive_file(y, cb, cr, o, h, out);
        return;
    }
    out.clear();
    out.reserve(static_cast<size_t>(o.width) * o.height * (o.color_type == PIXO_RGB ? 3 : 1) / 4 + 1024);
    write_headers(out, o, qt, h);
    BitSink sink(out);
    Emit e{sink, h};
    walk_scan(e, y, cb, cr, o);
    sink.align_with_ones();
    be16(out, 0xFFD9);
}

} // namespace pixo_host
