//! `pixo::jpeg`-compatible front end over the MI355X backend (C ABI: include/pixo_hip.h).
//!
//! Same names, fields, defaults and error variants as leerob/pixo v0.4.1
//! (`src/jpeg/mod.rs:88-447`, `src/color.rs:7-31`, `src/error.rs:6-91`), so that
//! `use pixo_hip as pixo;` is a drop-in for the JPEG path (`pixo::jpeg::{encode, encode_into, JpegOptions, ...}`,
//! `pixo::Error` with its `Display`) and for the PNG row-filter stage (`pixo::png::filter::apply_filters`).  NOT
//! compiled in this repository's image (no Rust toolchain); kept thin so that it can be reviewed by diff against
//! `src/error.rs:10-91`, `src/jpeg/mod.rs:121-447` and `src/png/filter.rs:51-62`.
#![allow(clippy::missing_safety_doc)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_int};

#[repr(u8)]
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum ColorType { Gray = 0, GrayAlpha = 1, Rgb = 2, Rgba = 3 }

impl ColorType {
    /// `src/color.rs:20-31`
    pub const fn bytes_per_pixel(self) -> usize {
        match self { ColorType::Gray => 1, ColorType::GrayAlpha => 2, ColorType::Rgb => 3, ColorType::Rgba => 4 }
    }
}

/// `pixo::Error` (`src/error.rs:10-48`): every variant, also those only the PNG encoder and the decoders raise, so
/// that `match` arms written against the reference keep compiling.
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum Error {
    InvalidDimensions { width: u32, height: u32 },
    InvalidDataLength { expected: usize, actual: usize },
    InvalidQuality(u8),
    InvalidCompressionLevel(u8),
    ImageTooLarge { width: u32, height: u32, max: u32 },
    UnsupportedColorType,
    CompressionError(String),
    InvalidRestartInterval(u16),
    InvalidDecode(String),
    UnsupportedDecode(String),
}
pub type Result<T> = std::result::Result<T, Error>;

/// `src/error.rs:50-91`, string for string (the C ABI's `pixo_hip_last_error` returns the same texts).
impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        match self {
            Error::InvalidDimensions { width, height } => write!(f, "Invalid image dimensions: {width}x{height}"),
            Error::InvalidDataLength { expected, actual } => write!(f, "Invalid pixel data length: expected {expected} bytes, got {actual}"),
            Error::InvalidQuality(q) => write!(f, "Invalid quality {q}: must be 1-100"),
            Error::InvalidCompressionLevel(level) => write!(f, "Invalid compression level {level}: must be 1-9"),
            Error::ImageTooLarge { width, height, max } => write!(f, "Image {width}x{height} exceeds maximum dimension {max}"),
            Error::UnsupportedColorType => write!(f, "Unsupported color type for this format"),
            Error::CompressionError(msg) => write!(f, "Compression error: {msg}"),
            Error::InvalidRestartInterval(interval) => write!(f, "Invalid restart interval {interval}: must be 1-65535 (or None to disable)"),
            Error::InvalidDecode(msg) => write!(f, "Decode error: {msg}"),
            Error::UnsupportedDecode(msg) => write!(f, "Unsupported: {msg}"),
        }
    }
}
impl std::error::Error for Error {}

#[repr(C)]
#[derive(Clone, Copy)]
struct COptions {
    width: u32, height: u32,
    color_type: u8, quality: u8, subsampling: u8, has_restart_interval: u8,
    restart_interval: u16,
    optimize_huffman: u8, progressive: u8, trellis_quant: u8,
}

const PIXO_ERR_BUFFER_TOO_SMALL: c_int = -9;

extern "C" {
    // include/pixo_hip.h
    fn pixo_hip_jpeg_encode_into(output: *mut u8, capacity: usize, data: *const u8, len: usize, opts: *const COptions,
                                 out_len: *mut usize) -> c_int;
    fn pixo_hip_jpeg_encode_multi(data: *const u8, len: usize, opts: *const COptions, devices: *const c_int, n_devices: u32,
                                  out: *mut *mut u8, out_len: *mut usize) -> c_int;
    fn pixo_hip_jpeg_encode_batch_multi(pixels: *const u8, opts: *const COptions, batch: u32, devices: *const c_int, n_devices: u32,
                                        arena: *mut u8, capacity: usize, offsets: *mut usize, lens: *mut usize) -> c_int;
    fn pixo_hip_encode_jpeg(data: *const u8, len: usize, width: u32, height: u32, color_type: u8, quality: u8, preset: u8,
                            subsampling_420: c_int, out: *mut *mut u8, out_len: *mut usize) -> c_int;
    fn pixo_hip_png_filter(data: *const u8, len: usize, width: u32, height: u32, bytes_per_pixel: u32, strategy: u8, flags: u32,
                           out: *mut u8, out_capacity: usize, adler32: *mut u32) -> c_int;
    fn pixo_hip_free(p: *mut u8);
    fn pixo_hip_copy_file(dst: *mut u8, src: *const u8, n: usize);
    fn pixo_hip_last_error() -> *const c_char;
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(pixo_hip_last_error()) }.to_string_lossy().into_owned()
}

pub mod jpeg {
    use super::*;

    #[derive(Debug, Clone, Copy, PartialEq, Eq)]
    pub enum Subsampling { S444, S420 }

    #[derive(Debug, Clone, Copy)]
    pub struct JpegOptions {
        pub width: u32, pub height: u32, pub color_type: ColorType, pub quality: u8,
        pub subsampling: Subsampling, pub restart_interval: Option<u16>,
        pub optimize_huffman: bool, pub progressive: bool, pub trellis_quant: bool,
    }
    impl Default for JpegOptions {
        fn default() -> Self {
            Self { width: 0, height: 0, color_type: ColorType::Rgb, quality: 75,
                   subsampling: Subsampling::S444, restart_interval: None,
                   optimize_huffman: false, progressive: false, trellis_quant: false }
        }
    }
    impl JpegOptions {
        pub fn fast(width: u32, height: u32, quality: u8) -> Self { Self { width, height, quality, ..Default::default() } }
        pub fn balanced(width: u32, height: u32, quality: u8) -> Self { Self { optimize_huffman: true, ..Self::fast(width, height, quality) } }
        pub fn max(width: u32, height: u32, quality: u8) -> Self {
            Self { subsampling: Subsampling::S420, optimize_huffman: true, progressive: true, trellis_quant: true, ..Self::fast(width, height, quality) }
        }
        pub fn from_preset(width: u32, height: u32, quality: u8, preset: u8) -> Self {
            match preset { 0 => Self::fast(width, height, quality), 2 => Self::max(width, height, quality), _ => Self::balanced(width, height, quality) }
        }
        pub fn builder(width: u32, height: u32) -> JpegOptionsBuilder { JpegOptionsBuilder::new(width, height) }
    }

    #[derive(Debug, Clone)]
    pub struct JpegOptionsBuilder { options: JpegOptions }
    impl JpegOptionsBuilder {
        pub fn new(width: u32, height: u32) -> Self { Self { options: JpegOptions { width, height, ..Default::default() } } }
        pub fn color_type(mut self, v: ColorType) -> Self { self.options.color_type = v; self }
        pub fn quality(mut self, v: u8) -> Self { self.options.quality = v; self }
        pub fn subsampling(mut self, v: Subsampling) -> Self { self.options.subsampling = v; self }
        pub fn restart_interval(mut self, v: Option<u16>) -> Self { self.options.restart_interval = v; self }
        pub fn optimize_huffman(mut self, v: bool) -> Self { self.options.optimize_huffman = v; self }
        pub fn progressive(mut self, v: bool) -> Self { self.options.progressive = v; self }
        pub fn trellis_quant(mut self, v: bool) -> Self { self.options.trellis_quant = v; self }
        pub fn preset(mut self, preset: u8) -> Self {
            let (w, h, c, q) = (self.options.width, self.options.height, self.options.color_type, self.options.quality);
            self.options = JpegOptions::from_preset(w, h, q, preset);
            self.options.color_type = c;
            self
        }
        #[must_use] pub fn build(self) -> JpegOptions { self.options }
    }

    fn to_c(o: &JpegOptions) -> COptions {
        COptions {
            width: o.width, height: o.height, color_type: o.color_type as u8, quality: o.quality,
            subsampling: matches!(o.subsampling, Subsampling::S420) as u8,
            has_restart_interval: o.restart_interval.is_some() as u8,
            restart_interval: o.restart_interval.unwrap_or(0),
            optimize_huffman: o.optimize_huffman as u8, progressive: o.progressive as u8,
            trellis_quant: o.trellis_quant as u8,
        }
    }

    fn error_from(status: c_int, o: &JpegOptions, len: usize) -> Error {
        let msg = last_error();
        match status {
            -1 => Error::InvalidDimensions { width: o.width, height: o.height },
            -2 => {
                Error::InvalidDataLength { expected: o.width as usize * o.height as usize * o.color_type.bytes_per_pixel(), actual: len }
            }
            -3 => Error::InvalidQuality(o.quality),
            -4 => Error::ImageTooLarge { width: o.width, height: o.height, max: 65535 },
            -5 => Error::UnsupportedColorType,
            -7 => Error::InvalidRestartInterval(o.restart_interval.unwrap_or(0)),
            _ => Error::CompressionError(msg.trim_start_matches("Compression error: ").to_string()),
        }
    }

    /// `pixo::jpeg::encode_into` (reference src/jpeg/mod.rs:328): clears `output` and writes the file into it, reusing
    /// its allocation.  The library writes straight into the vector's spare capacity; when the file does not fit it says
    /// how many bytes it needs (PIXO_ERR_BUFFER_TOO_SMALL leaves the buffer untouched) and the call is repeated once —
    /// the reserve-and-retry protocol `pixo_hip_jpeg_encode_into` exists for.  Like the reference, validation errors are
    /// returned before `output` is touched.
    pub fn encode_into(output: &mut Vec<u8>, data: &[u8], options: &JpegOptions) -> Result<()> {
        let c = to_c(options);
        // the reference reserves data.len() / 4 (src/jpeg/mod.rs:375); keep whatever the caller's vector already has
        let mut capacity = output.capacity().max(data.len() / 4);
        for _ in 0..2 {
            // grow the caller's vector in place, then hand the library the raw pointer / capacity pair — no second `&mut` to
            // the vector is alive across the FFI call.  Its length is not changed before success: a validation error
            // (which the library returns before it writes anything) leaves `output` as it was, like the reference.
            if output.capacity() < capacity {
                output.reserve_exact(capacity - output.len());
            }
            let (ptr, cap) = (output.as_mut_ptr(), output.capacity());
            let mut needed = 0usize;
            let rc = unsafe { pixo_hip_jpeg_encode_into(ptr, cap, data.as_ptr(), data.len(), &c, &mut needed) };
            if rc == 0 {
                // SAFETY: the library has initialised `needed` <= `cap` bytes behind `ptr`
                unsafe { output.set_len(needed) };
                return Ok(());
            }
            if rc != PIXO_ERR_BUFFER_TOO_SMALL {
                return Err(error_from(rc, options, data.len()));
            }
            capacity = needed;
        }
        Err(Error::CompressionError("output size changed between two identical calls".to_string()))
    }

    /// `pixo::jpeg::encode` (reference src/jpeg/mod.rs:88).
    #[must_use = "encoding produces a JPEG file that should be used"]
    pub fn encode(data: &[u8], options: &JpegOptions) -> Result<Vec<u8>> {
        let mut out = Vec::new();
        encode_into(&mut out, data, options)?;
        Ok(out)
    }

    /// Not in the reference: `encode` with the image's MCU-row bands spread over several GPUs of this process
    /// (`pixo_hip_jpeg_encode_multi`; config 4: one 16384 x 16384 image over 8 MI355X).  Same bytes as `encode`.
    pub fn encode_on_devices(data: &[u8], options: &JpegOptions, devices: &[i32]) -> Result<Vec<u8>> {
        let c = to_c(options);
        let (mut p, mut n) = (std::ptr::null_mut::<u8>(), 0usize);
        let rc = unsafe { pixo_hip_jpeg_encode_multi(data.as_ptr(), data.len(), &c, devices.as_ptr(), devices.len() as u32, &mut p, &mut n) };
        if rc != 0 { return Err(error_from(rc, options, data.len())); }
        // into the Vec by the library's copy threads (a 178 MB file: 3 ms instead of the 28 of a one-thread copy into
        // fresh pages), then the library's block goes back to its cache
        let mut out = Vec::<u8>::with_capacity(n);
        unsafe {
            pixo_hip_copy_file(out.as_mut_ptr(), p, n);
            out.set_len(n);
            pixo_hip_free(p);
        }
        Ok(out)
    }

    /// Not in the reference (its counterpart is a loop over `encode`, src/jpeg/mod.rs:88): `images.len() / image_bytes` equally
    /// sized images, back to back in host memory, encoded by several GPUs of this process (`pixo_hip_jpeg_encode_batch_multi`;
    /// configs[2] on a node: every GPU fetches its share over its own PCIe link, encodes it in one pass and copies its files to
    /// their final place).  Returns the files back to back and where each begins: file i = `arena[offsets[i]..offsets[i + 1]]`.
    pub fn encode_batch_on_devices(images: &[u8], options: &JpegOptions, batch: u32, devices: &[i32]) -> Result<(Vec<u8>, Vec<usize>)> {
        let c = to_c(options);
        let n = batch as usize;
        // The C entry takes no length (it reads batch * image_bytes from the pointer): check here what `encode` lets the C side
        // check, so that a safe caller cannot make it read beyond the slice.
        let image_bytes = options.width as usize * options.height as usize * options.color_type.bytes_per_pixel();
        if images.len() != n * image_bytes {
            return Err(Error::InvalidDataLength { expected: n * image_bytes, actual: images.len() });
        }
        if devices.is_empty() {
            return Err(Error::CompressionError("encode_batch_on_devices: no device listed".to_string()));
        }
        let (mut offsets, mut lens) = (vec![0usize; n], vec![0usize; n]);
        let mut arena = Vec::<u8>::with_capacity(images.len() / 4 + 4096);
        for _ in 0..2 {
            let rc = unsafe {
                pixo_hip_jpeg_encode_batch_multi(images.as_ptr(), &c, batch, devices.as_ptr(), devices.len() as u32, arena.as_mut_ptr(),
                                                 arena.capacity(), offsets.as_mut_ptr(), lens.as_mut_ptr())
            };
            let total = if n > 0 { offsets[n - 1] + lens[n - 1] } else { 0 };
            if rc == 0 {
                // SAFETY: the library has initialised `total` <= capacity bytes of the arena
                unsafe { arena.set_len(total) };
                offsets.push(total);
                return Ok((arena, offsets));
            }
            if rc != PIXO_ERR_BUFFER_TOO_SMALL { return Err(error_from(rc, options, images.len())); }
            arena.reserve_exact(total); // (offsets / lens were filled in: the second attempt fits)
        }
        Err(Error::CompressionError("output size changed between two identical calls".to_string()))
    }
}

/// The reference's flat wasm export (`src/wasm.rs:113-142`: `encode_jpeg(data, width, height, color_type, quality, preset,
/// subsampling_420)`), same seven arguments, through `pixo_hip_encode_jpeg`; errors as the reference's strings.
pub fn encode_jpeg(data: &[u8], width: u32, height: u32, color_type: u8, quality: u8, preset: u8, subsampling_420: bool)
    -> std::result::Result<Vec<u8>, String> {
    let (mut p, mut n) = (std::ptr::null_mut::<u8>(), 0usize);
    let rc = unsafe { pixo_hip_encode_jpeg(data.as_ptr(), data.len(), width, height, color_type, quality, preset, subsampling_420 as c_int, &mut p, &mut n) };
    if rc != 0 { return Err(last_error()); }
    let mut out = Vec::<u8>::with_capacity(n);
    unsafe {
        pixo_hip_copy_file(out.as_mut_ptr(), p, n);
        out.set_len(n);
        pixo_hip_free(p);
    }
    Ok(out)
}

/// `pixo::png` — the row-filter stage of config 5 (`src/png/filter.rs:51-206`) and the Adler-32 of its output.
pub mod png {
    use super::*;

    /// `src/png/mod.rs:345-364`, declaration order = the C ABI's `pixo_png_filter_strategy`.
    #[derive(Debug, Clone, Copy, PartialEq, Eq)]
    pub enum FilterStrategy { None, Sub, Up, Average, Paeth, MinSum, Adaptive, AdaptiveFast, Bigrams }

    pub mod filter {
        use super::*;

        /// `pixo::png::filter::apply_filters(data, width, height, bytes_per_pixel, &options)` with the strategy taken out
        /// of `PngOptions` (the only field the function reads, `src/png/filter.rs:71`): one filter-type byte + the
        /// filtered row per image row, exactly what the reference hands to its DEFLATE.  The reference's signature
        /// is infallible; device failures therefore panic, like an allocation failure would.
        pub fn apply_filters(data: &[u8], width: u32, height: u32, bytes_per_pixel: usize, strategy: FilterStrategy) -> Vec<u8> {
            apply_filters_with_adler32(data, width, height, bytes_per_pixel, strategy).0
        }

        /// The same, plus the zlib Adler-32 of the filtered stream (`src/simd/fallback.rs:8-25`, computed by the
        /// reference inside its zlib wrapper, `src/compress/deflate.rs:1044`): lets the caller skip that pass.
        pub fn apply_filters_with_adler32(data: &[u8], width: u32, height: u32, bytes_per_pixel: usize, strategy: FilterStrategy) -> (Vec<u8>, u32) {
            let n = height as usize * (width as usize * bytes_per_pixel + 1);
            let mut out = Vec::<u8>::with_capacity(n);
            let mut adler = 0u32;
            let rc = unsafe {
                pixo_hip_png_filter(data.as_ptr(), data.len(), width, height, bytes_per_pixel as u32, strategy as u8, 0,
                                    out.as_mut_ptr(), n, &mut adler)
            };
            assert!(rc == 0, "pixo_hip_png_filter: {}", last_error());
            unsafe { out.set_len(n) };
            (out, adler)
        }
    }
}
