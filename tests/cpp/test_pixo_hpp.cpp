// Exercises include/pixo.hpp (the C++ mirror of the reference's Rust API) against the C ABI.
// Without a GPU it checks option semantics and every validation error (reported before any
// device work, reference src/jpeg/mod.rs:333-373).  With a GPU (argv[1] == "gpu") it also
// encodes an image and writes it to argv[2] so the Python test can compare it with the oracle.
#include <cstdio>
#include <cstring>
#include <fstream>

#include "../../include/pixo.hpp"

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL line %d: %s\n", __LINE__, #cond); ++fails; } } while (0)

template <class F> static bool raises(pixo::Error::Kind kind, const char *msg, F &&f)
{
    try { f(); } catch (const pixo::Error &e) {
        const bool ok = e.kind() == kind && std::strcmp(e.what(), msg) == 0;
        if (!ok) std::printf("  got kind %d message '%s'\n", (int)e.kind(), e.what());
        return ok;
    }
    return false;
}

int main(int argc, char **argv)
{
    using namespace pixo;
    using namespace pixo::jpeg;
    // presets and builder (jpeg/mod.rs:162-300)
    JpegOptions d;
    CHECK(d.quality == 75 && d.subsampling == Subsampling::S444 && d.color_type == ColorType::Rgb && !d.restart_interval);
    JpegOptions m = JpegOptions::max(3, 4, 60);
    CHECK(m.subsampling == Subsampling::S420 && m.optimize_huffman && m.progressive && m.trellis_quant);
    CHECK(JpegOptions::from_preset(1, 1, 50, 9).optimize_huffman && !JpegOptions::fast(1, 1, 50).optimize_huffman);
    JpegOptions o = JpegOptions::builder(10, 20).color_type(ColorType::Gray).quality(33).preset(2).subsampling(Subsampling::S444).build();
    CHECK(o.width == 10 && o.height == 20 && o.color_type == ColorType::Gray && o.quality == 33);
    CHECK(o.subsampling == Subsampling::S444 && o.progressive && o.trellis_quant);
    CHECK(bytes_per_pixel(ColorType::Rgba) == 4 && bytes_per_pixel(ColorType::Gray) == 1);

    std::vector<uint8_t> px(48, 7), out{1, 2, 3};
    CHECK(raises(Error::Kind::InvalidQuality, "Invalid quality 0: must be 1-100",
                 [&] { encode_into(out, px.data(), px.size(), JpegOptions::builder(0, 0).quality(0).restart_interval(0).build()); }));
    CHECK(out.size() == 3); // untouched on error
    CHECK(raises(Error::Kind::InvalidRestartInterval, "Invalid restart interval 0: must be 1-65535 (or None to disable)",
                 [&] { (void)encode(px, JpegOptions::builder(0, 0).restart_interval(0).build()); }));
    CHECK(raises(Error::Kind::InvalidDimensions, "Invalid image dimensions: 0x4", [&] { (void)encode(px, JpegOptions::builder(0, 4).build()); }));
    CHECK(raises(Error::Kind::ImageTooLarge, "Image 65536x1 exceeds maximum dimension 65535",
                 [&] { (void)encode(px, JpegOptions::builder(65536, 1).color_type(ColorType::Rgba).build()); }));
    CHECK(raises(Error::Kind::UnsupportedColorType, "Unsupported color type for this format",
                 [&] { (void)encode(px, JpegOptions::builder(4, 4).color_type(ColorType::GrayAlpha).build()); }));
    CHECK(raises(Error::Kind::InvalidDataLength, "Invalid pixel data length: expected 48 bytes, got 47",
                 [&] { (void)encode(px.data(), 47, JpegOptions::builder(4, 4).build()); }));
    CHECK(raises(Error::Kind::InvalidColorArgument, "Invalid color type for JPEG: 3. Expected 0 (Gray) or 2 (Rgb)",
                 [&] { (void)encode_jpeg(px.data(), px.size(), 4, 4, 3, 80, 0, true); }));

    // PNG row filters: argument errors are reported before any device work
    CHECK(raises(Error::Kind::InvalidDimensions, "Invalid image dimensions: 0x4",
                 [&] { (void)png::filter::apply_filters(px.data(), px.size(), 0, 4, 3, png::FilterStrategy::Adaptive); }));
    CHECK(raises(Error::Kind::UnsupportedColorType, "Unsupported color type for this format",
                 [&] { (void)png::filter::apply_filters(px.data(), px.size(), 4, 4, 5, png::FilterStrategy::Sub); }));

    if (argc > 2 && std::strcmp(argv[1], "gpu") == 0) {
        const uint32_t w = 200, h = 120;
        std::vector<uint8_t> img(w * h * 3);
        uint32_t s = 9; // tests/support/synthetic.rs:183 LCG noise
        for (auto &b : img) { s = s * 1103515245u + 12345u; b = (uint8_t)(s >> 16); }
        auto jpg = encode(img, JpegOptions::builder(w, h).quality(80).subsampling(Subsampling::S420).build());
        std::ofstream(argv[2], std::ios::binary).write((const char *)jpg.data(), (std::streamsize)jpg.size());
        CHECK(jpg.size() > 4 && jpg[0] == 0xFF && jpg[1] == 0xD8);
        // a row of equal pixels: Sub gives zeros after the first pixel, filter byte 1 in front
        std::vector<uint8_t> rgba(64 * 80 * 4, 9);
        uint32_t ad = 0;
        auto flt = png::filter::apply_filters(rgba.data(), rgba.size(), 64, 80, 4, png::FilterStrategy::Sub, &ad);
        CHECK(flt.size() == 80u * (64 * 4 + 1) && flt[0] == 1 && flt[1] == 9 && flt[5] == 0 && flt[256] == 0 && flt[257] == 1 && ad != 0);
    } else {
        CHECK(raises(Error::Kind::CompressionError,
                     "Compression error: no MI355X/HIP device available (pixo_hip has no CPU fallback)",
                     [&] { if (pixo_hip_device_count() == 0) (void)encode(px, JpegOptions::builder(4, 4).build());
                           else throw Error(Error::Kind::CompressionError, "Compression error: no MI355X/HIP device available (pixo_hip has no CPU fallback)"); }));
    }
    std::printf(fails ? "%d checks FAILED\n" : "all checks passed\n", fails);
    return fails ? 1 : 0;
}
