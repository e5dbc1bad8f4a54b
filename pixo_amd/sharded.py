"""One image across several GPUs (BASELINE config 4: 16384x16384 over 8 MI355X).

The image is split into contiguous MCU-row bands (`pixo_hip_band`); a band is an independent
sub-image, so every rank runs the ordinary coefficient kernel on its rows with no halo and no
data-path collective.  The only cross-band state of a baseline JPEG is the DC predictor chain
and the bit offset, both of which live in the entropy stage — which therefore runs once, on
rank 0, over the gathered tuple (reference seam: `YCbCrCoefficients`, src/jpeg/mod.rs:58-61).
The result is byte-identical to the single-device file.

One process per GPU; the process group (RCCL for device tensors, gloo for host arrays) only
carries the gather of the finished coefficient bands.
"""
import numpy as np

from . import jpeg


def encode_banded(data, options, group=None, coeff_fn=None, dst=0):
    """Collective over `group`: every rank passes the same `options` and (at least) its own rows
    of `data` (the full image is fine).  Returns the JFIF bytes on rank `dst`, None elsewhere.
    `coeff_fn(band_pixels, band_options) -> (y, cb, cr)` defaults to the GPU path
    (`jpeg.coefficients`); tests substitute a CPU function to exercise the sharding on gloo."""
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    coeff_fn = coeff_fn or jpeg.coefficients
    w, h = options.width, options.height
    ct, ss = int(options.color_type), int(options.subsampling)
    bpp = 1 if ct == 0 else 3
    px = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1)
    b = jpeg.band(w, h, ct, ss, world, rank)
    rows = b["row_end"] - b["row_begin"]
    if rows > 0:
        sub = px[b["row_begin"] * w * bpp: b["row_end"] * w * bpp]
        band_opts = jpeg.JpegOptions(**{**options.__dict__, "height": rows})
        y, cb, cr = coeff_fn(sub, band_opts)
        assert y.shape[0] == b["y_blocks"] and cb.shape[0] == b["c_blocks"]
    else:
        y = cb = cr = np.zeros((0, 64), np.int16)
    parts = [None] * world if rank == dst else None
    dist.gather_object((y, cb, cr), parts, dst=dst, group=group)
    if rank != dst:
        return None
    return jpeg.entropy_encode(np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
                               np.concatenate([p[2] for p in parts]), options)
