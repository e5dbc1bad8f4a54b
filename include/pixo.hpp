// pixo.hpp — C++17 host-side mirror of the reference's Rust API for the JPEG path, header-only
// over the C ABI of pixo_hip.h.  (The reference is a Rust crate and there is no Rust toolchain
// in the build image; the host side above the C ABI is therefore written in C++ with the same
// names, argument meaning and error behaviour, so that code and tests read like the
// reference's.  The Rust binding a pixo maintainer would add is shown in INTEGRATION.md and
// sketched under rust/.)
//
//   reference (leerob/pixo v0.4.1)                      here
//   pixo::ColorType                    src/color.rs:9   pixo::ColorType
//   pixo::Error / pixo::Result<T>      src/error.rs:6   pixo::Error (exception), values returned
//   pixo::jpeg::Subsampling            jpeg/mod.rs:96   pixo::jpeg::Subsampling
//   pixo::jpeg::JpegOptions + presets  jpeg/mod.rs:121  pixo::jpeg::JpegOptions
//   pixo::jpeg::JpegOptionsBuilder     jpeg/mod.rs:230  pixo::jpeg::JpegOptionsBuilder
//   pixo::jpeg::encode                 jpeg/mod.rs:88   pixo::jpeg::encode
//   pixo::jpeg::encode_into            jpeg/mod.rs:328  pixo::jpeg::encode_into
//   (wasm) encode_jpeg                 wasm.rs:113      pixo::encode_jpeg
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "pixo_hip.h"

namespace pixo {

enum class ColorType : uint8_t { Gray = 0, GrayAlpha = 1, Rgb = 2, Rgba = 3 };

constexpr size_t bytes_per_pixel(ColorType c)
{
    return c == ColorType::Gray ? 1 : c == ColorType::GrayAlpha ? 2 : c == ColorType::Rgb ? 3 : 4;
}

// pixo::Error: one kind per variant this path can raise; what() is the reference's Display text.
class Error : public std::runtime_error {
  public:
    enum class Kind {
        InvalidDimensions, InvalidDataLength, InvalidQuality, ImageTooLarge, UnsupportedColorType,
        CompressionError, InvalidRestartInterval, InvalidColorArgument, BufferTooSmall, Unknown
    };
    Error(Kind k, const std::string &msg) : std::runtime_error(msg), kind_(k) {}
    Kind kind() const { return kind_; }
    static Error from_status(int status)
    {
        Kind k = Kind::Unknown;
        switch (status) {
        case PIXO_ERR_INVALID_DIMENSIONS: k = Kind::InvalidDimensions; break;
        case PIXO_ERR_INVALID_DATA_LENGTH: k = Kind::InvalidDataLength; break;
        case PIXO_ERR_INVALID_QUALITY: k = Kind::InvalidQuality; break;
        case PIXO_ERR_IMAGE_TOO_LARGE: k = Kind::ImageTooLarge; break;
        case PIXO_ERR_UNSUPPORTED_COLOR_TYPE: k = Kind::UnsupportedColorType; break;
        case PIXO_ERR_COMPRESSION: k = Kind::CompressionError; break;
        case PIXO_ERR_INVALID_RESTART_INTERVAL: k = Kind::InvalidRestartInterval; break;
        case PIXO_ERR_INVALID_COLOR_ARG: k = Kind::InvalidColorArgument; break;
        case PIXO_ERR_BUFFER_TOO_SMALL: k = Kind::BufferTooSmall; break;
        default: break;
        }
        return Error(k, pixo_hip_last_error());
    }

  private:
    Kind kind_;
};

namespace jpeg {

enum class Subsampling : uint8_t { S444 = 0, S420 = 1 };

class JpegOptionsBuilder;

// jpeg/mod.rs:121-157; Default = quality 75, 4:4:4, RGB, no restarts, width/height 0.
struct JpegOptions {
    uint32_t width = 0;
    uint32_t height = 0;
    ColorType color_type = ColorType::Rgb;
    uint8_t quality = 75;
    Subsampling subsampling = Subsampling::S444;
    std::optional<uint16_t> restart_interval;
    bool optimize_huffman = false;
    bool progressive = false;
    bool trellis_quant = false;

    static JpegOptions fast(uint32_t w, uint32_t h, uint8_t q) { return from_preset(w, h, q, 0); }
    static JpegOptions balanced(uint32_t w, uint32_t h, uint8_t q) { return from_preset(w, h, q, 1); }
    static JpegOptions max(uint32_t w, uint32_t h, uint8_t q) { return from_preset(w, h, q, 2); }
    static JpegOptions from_preset(uint32_t w, uint32_t h, uint8_t q, uint8_t preset)
    { // jpeg/mod.rs:162-216
        pixo_jpeg_options c;
        pixo_jpeg_options_from_preset(&c, w, h, q, preset);
        JpegOptions o;
        o.width = c.width; o.height = c.height; o.quality = c.quality;
        o.color_type = static_cast<ColorType>(c.color_type);
        o.subsampling = static_cast<Subsampling>(c.subsampling);
        o.optimize_huffman = c.optimize_huffman; o.progressive = c.progressive; o.trellis_quant = c.trellis_quant;
        return o;
    }
    static JpegOptionsBuilder builder(uint32_t width, uint32_t height);

    pixo_jpeg_options to_c() const
    {
        pixo_jpeg_options c{};
        c.width = width; c.height = height;
        c.color_type = static_cast<uint8_t>(color_type); c.quality = quality;
        c.subsampling = static_cast<uint8_t>(subsampling);
        c.has_restart_interval = restart_interval.has_value();
        c.restart_interval = restart_interval.value_or(0);
        c.optimize_huffman = optimize_huffman; c.progressive = progressive; c.trellis_quant = trellis_quant;
        return c;
    }
};

// jpeg/mod.rs:230-300
class JpegOptionsBuilder {
  public:
    JpegOptionsBuilder(uint32_t width, uint32_t height) { o_.width = width; o_.height = height; }
    JpegOptionsBuilder &color_type(ColorType v) { o_.color_type = v; return *this; }
    JpegOptionsBuilder &quality(uint8_t v) { o_.quality = v; return *this; }
    JpegOptionsBuilder &subsampling(Subsampling v) { o_.subsampling = v; return *this; }
    JpegOptionsBuilder &restart_interval(std::optional<uint16_t> v) { o_.restart_interval = v; return *this; }
    JpegOptionsBuilder &optimize_huffman(bool v) { o_.optimize_huffman = v; return *this; }
    JpegOptionsBuilder &progressive(bool v) { o_.progressive = v; return *this; }
    JpegOptionsBuilder &trellis_quant(bool v) { o_.trellis_quant = v; return *this; }
    // keeps width, height, colour type and quality (jpeg/mod.rs:285-293)
    JpegOptionsBuilder &preset(uint8_t p)
    {
        const ColorType keep = o_.color_type;
        o_ = JpegOptions::from_preset(o_.width, o_.height, o_.quality, p);
        o_.color_type = keep;
        return *this;
    }
    JpegOptions build() const { return o_; }

  private:
    JpegOptions o_;
};

inline JpegOptionsBuilder JpegOptions::builder(uint32_t width, uint32_t height)
{
    return JpegOptionsBuilder(width, height);
}

// pixo::jpeg::encode_into (jpeg/mod.rs:328): clears and refills `output`; throws pixo::Error and
// leaves `output` untouched on failure (validation precedes output.clear() in the reference).
inline void encode_into(std::vector<uint8_t> &output, const uint8_t *data, size_t len, const JpegOptions &options)
{
    const pixo_jpeg_options c = options.to_c();
    uint8_t *buf = nullptr;
    size_t n = 0;
    const int rc = pixo_hip_jpeg_encode(data, len, &c, &buf, &n);
    if (rc != PIXO_OK) throw Error::from_status(rc);
    output.assign(buf, buf + n);
    pixo_hip_free(buf);
}

// pixo::jpeg::encode (jpeg/mod.rs:88)
[[nodiscard]] inline std::vector<uint8_t> encode(const uint8_t *data, size_t len, const JpegOptions &options)
{
    std::vector<uint8_t> out;
    encode_into(out, data, len, options);
    return out;
}
[[nodiscard]] inline std::vector<uint8_t> encode(const std::vector<uint8_t> &data, const JpegOptions &options)
{
    return encode(data.data(), data.size(), options);
}

} // namespace jpeg

namespace png {
// pixo::png::FilterStrategy in declaration order (src/png/mod.rs:345-364)
enum class FilterStrategy : uint8_t { None = 0, Sub, Up, Average, Paeth, MinSum, Adaptive, AdaptiveFast, Bigrams };

namespace filter {
// pixo::png::filter::apply_filters (src/png/filter.rs:51-206): one filter-type byte + the filtered row for every
// row — what the reference hands to its DEFLATE.  `adler32`, when given, receives the zlib checksum of the
// returned bytes (src/simd/fallback.rs:8-25, computed once over the whole stream: src/compress/deflate.rs:1044).
[[nodiscard]] inline std::vector<uint8_t> apply_filters(const uint8_t *data, size_t len, uint32_t width, uint32_t height,
                                                        size_t bytes_per_pixel, FilterStrategy strategy,
                                                        uint32_t *adler32 = nullptr, uint32_t flags = 0)
{
    std::vector<uint8_t> out((size_t)height * ((size_t)width * bytes_per_pixel + 1));
    uint32_t ad = 0;
    const int rc = pixo_hip_png_filter(data, len, width, height, (uint32_t)bytes_per_pixel, (uint8_t)strategy, flags,
                                       out.data(), out.size(), &ad);
    if (rc != PIXO_OK) throw Error::from_status(rc);
    if (adler32) *adler32 = ad;
    return out;
}
} // namespace filter
} // namespace png

// The reference's flat wasm export (src/wasm.rs:113-142), same seven arguments.
[[nodiscard]] inline std::vector<uint8_t> encode_jpeg(const uint8_t *data, size_t len, uint32_t width, uint32_t height,
                                                      uint8_t color_type, uint8_t quality, uint8_t preset,
                                                      bool subsampling_420)
{
    uint8_t *buf = nullptr;
    size_t n = 0;
    const int rc = pixo_hip_encode_jpeg(data, len, width, height, color_type, quality, preset, subsampling_420, &buf, &n);
    if (rc != PIXO_OK) throw Error::from_status(rc);
    std::vector<uint8_t> out(buf, buf + n);
    pixo_hip_free(buf);
    return out;
}

} // namespace pixo
