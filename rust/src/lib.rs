//! `pixo::jpeg`-compatible front end over the MI355X backend (C ABI: include/pixo_hip.h).
//!
//! Same names, fields, defaults and error variants as leerob/pixo v0.4.1
//! (`src/jpeg/mod.rs:88-447`, `src/color.rs:7-31`, `src/error.rs:6-91`), so that
//! `use pixo_hip as pixo;` is a drop-in for the baseline JPEG path.  NOT compiled in this
//! repository's image (no Rust toolchain); kept thin so that it can be reviewed by diff.
#![allow(clippy::missing_safety_doc)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_int};

#[repr(u8)]
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum ColorType { Gray = 0, GrayAlpha = 1, Rgb = 2, Rgba = 3 }

#[derive(Debug, Clone, PartialEq, Eq)]
pub enum Error {
    InvalidDimensions { width: u32, height: u32 },
    InvalidDataLength { expected: usize, actual: usize },
    InvalidQuality(u8),
    ImageTooLarge { width: u32, height: u32, max: u32 },
    UnsupportedColorType,
    CompressionError(String),
    InvalidRestartInterval(u16),
}
pub type Result<T> = std::result::Result<T, Error>;

#[repr(C)]
#[derive(Clone, Copy)]
struct COptions {
    width: u32, height: u32,
    color_type: u8, quality: u8, subsampling: u8, has_restart_interval: u8,
    restart_interval: u16,
    optimize_huffman: u8, progressive: u8, trellis_quant: u8,
}

extern "C" {
    fn pixo_hip_jpeg_encode(data: *const u8, len: usize, opts: *const COptions,
                            out: *mut *mut u8, out_len: *mut usize) -> c_int;
    fn pixo_hip_free(p: *mut u8);
    fn pixo_hip_last_error() -> *const c_char;
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
        let msg = unsafe { CStr::from_ptr(pixo_hip_last_error()) }.to_string_lossy().into_owned();
        match status {
            -1 => Error::InvalidDimensions { width: o.width, height: o.height },
            -2 => {
                let bpp = if o.color_type == ColorType::Rgb { 3 } else { 1 };
                Error::InvalidDataLength { expected: o.width as usize * o.height as usize * bpp, actual: len }
            }
            -3 => Error::InvalidQuality(o.quality),
            -4 => Error::ImageTooLarge { width: o.width, height: o.height, max: 65535 },
            -5 => Error::UnsupportedColorType,
            -7 => Error::InvalidRestartInterval(o.restart_interval.unwrap_or(0)),
            _ => Error::CompressionError(msg.trim_start_matches("Compression error: ").to_string()),
        }
    }

    /// `pixo::jpeg::encode_into` (reference src/jpeg/mod.rs:328).
    pub fn encode_into(output: &mut Vec<u8>, data: &[u8], options: &JpegOptions) -> Result<()> {
        let c = to_c(options);
        let (mut p, mut n) = (std::ptr::null_mut::<u8>(), 0usize);
        let rc = unsafe { pixo_hip_jpeg_encode(data.as_ptr(), data.len(), &c, &mut p, &mut n) };
        if rc != 0 { return Err(error_from(rc, options, data.len())); }
        output.clear();
        output.extend_from_slice(unsafe { std::slice::from_raw_parts(p, n) });
        unsafe { pixo_hip_free(p) };
        Ok(())
    }

    /// `pixo::jpeg::encode` (reference src/jpeg/mod.rs:88).
    #[must_use = "encoding produces a JPEG file that should be used"]
    pub fn encode(data: &[u8], options: &JpegOptions) -> Result<Vec<u8>> {
        let mut out = Vec::new();
        encode_into(&mut out, data, options)?;
        Ok(out)
    }
}
