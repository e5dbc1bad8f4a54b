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


def encode_banded_device(d_band_pixels, options, group=None, dst=0, coeff_fn=None, entropy_fn=None):
    """Device-resident form of `encode_banded`: every rank holds ITS band of the image
    (`jpeg.band(..., world, rank)` rows, tightly packed) as a torch uint8 tensor on its own GPU.
    The rank's coefficient kernel fills a device tuple; the finished bands travel to rank `dst`
    with ONE collective per plane (`torch.distributed.gather` — RCCL over xGMI for device tensors:
    7 peers send 1/8 of the tuple each, point to point); rank `dst` runs the device entropy stage
    over the stitched tuple.  No host copy of coefficients anywhere.  Returns the file on `dst`.

    `coeff_fn(d_pixels, band_options, y, cb, cr)` / `entropy_fn(y, cb, cr, options)` default to
    the GPU kernels; tests substitute CPU functions to run the same code over gloo."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    w, h = options.width, options.height
    ct, ss = int(options.color_type), int(options.subsampling)
    bands = [jpeg.band(w, h, ct, ss, world, r) for r in range(world)]
    b = bands[rank]
    rows = b["row_end"] - b["row_begin"]
    ymax = max(max(x["y_blocks"] for x in bands), 1)
    cmax = max(max(x["c_blocks"] for x in bands), 1)
    dev = d_band_pixels.device
    # equal-sized send buffers (gather needs them): bands differ by at most one MCU row
    y = torch.zeros((ymax, 64), dtype=torch.int16, device=dev)
    cb = torch.zeros((cmax, 64), dtype=torch.int16, device=dev)
    cr = torch.zeros((cmax, 64), dtype=torch.int16, device=dev)
    if rows > 0:
        band_opts = jpeg.JpegOptions(**{**options.__dict__, "height": rows})
        if coeff_fn is not None:
            coeff_fn(d_band_pixels, band_opts, y, cb, cr)
        else:
            jpeg.coefficients_device(d_band_pixels, w, rows, ct, ss, options.quality, y, cb, cr,
                                     stream=torch.cuda.current_stream(dev).cuda_stream)
    outs = []
    for t in (y, cb, cr):
        tb = t.view(torch.uint8)  # neither RCCL nor gloo has a 16-bit integer type: move bytes
        parts = [torch.empty_like(tb) for _ in range(world)] if rank == dst else None
        dist.gather(tb, parts, dst=dst, group=group)
        outs.append([p.view(torch.int16) for p in parts] if parts is not None else None)
    if rank != dst:
        return None
    fy = torch.cat([outs[0][r][: bands[r]["y_blocks"]] for r in range(world)])
    fcb = torch.cat([outs[1][r][: bands[r]["c_blocks"]] for r in range(world)])
    fcr = torch.cat([outs[2][r][: bands[r]["c_blocks"]] for r in range(world)])
    if entropy_fn is not None:
        return entropy_fn(fy, fcb, fcr, options)
    torch.cuda.synchronize(dev)  # the entropy stage runs on the library's own stream
    if fcb.shape[0] == 0:  # gray: the planes are unused but must be valid pointers
        fcb = fcr = fy
    return jpeg.entropy_encode_device(fy, fcb, fcr, options)
