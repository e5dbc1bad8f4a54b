"""pixo_amd — MI355X (gfx950) backend for pixo's JPEG encode path.

Host-side mirror of the reference's public API for this path:

    from pixo_amd import ColorType, jpeg
    opts = jpeg.JpegOptions.builder(w, h).color_type(ColorType.Rgb).quality(80) \
               .subsampling(jpeg.Subsampling.S420).build()
    data = jpeg.encode(pixels, opts)

Everything below this package is the C ABI of include/pixo_hip.h (pixo_amd/libpixo_hip.so).
"""
from . import error, jpeg, png  # noqa: F401
from .color import ColorType  # noqa: F401
from .error import Error  # noqa: F401

__all__ = ["ColorType", "Error", "error", "jpeg"]
