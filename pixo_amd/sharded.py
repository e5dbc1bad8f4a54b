"""One image across several GPUs (BASELINE config 4: 16384x16384 over 8 MI355X), one process per GPU.

The image is split into contiguous MCU-row bands (`pixo_hip_band`); a band is an independent
sub-image for everything up to the quantised coefficients.  The only cross-band state of a baseline
JPEG is the DC predictor chain and the bit position of the scan (src/jpeg/mod.rs:1417-1419,
src/bits.rs:216-272), so every rank also ENTROPY-CODES its own band (`jpeg.BandEncoder`) and the ranks
exchange, per band:

    1. the last quantised DC of Y, Cb, Cr        3 x i16   all_gather
   (1b. with optimize_huffman: its symbol counts   536 x u64 all_reduce(sum))
    2. its length in bits                          1 x u64   all_gather

— no coefficient ever leaves its GPU.  Each rank packs and 0xFF-stuffs its band at its global bit
offset; the finished pieces (the compressed bytes, ~1/5 of the coefficient bytes for noise, far less for
photographs) go to rank `dst`, which writes the headers and splices.  Byte-identical to the
single-device file.

Progressive scans and restart markers are not band-codable (one DC chain per scan / byte-aligned
segments that ignore band boundaries): for those the finished coefficient bands are gathered on `dst`
(`encode_gathered*`), which runs the ordinary entropy stage over the stitched tuple.

The process group (RCCL for device tensors, gloo for host arrays) only carries these exchanges.
"""
import numpy as np

from . import jpeg


def band_codable(options) -> bool:
    """Baseline scan without restart markers (what `pixo_hip_band_encoder_create` accepts)."""
    units = None
    if options.restart_interval is not None:
        w, h = options.width, options.height
        unit = 16 if (int(options.color_type) != 0 and int(options.subsampling) == 1) else 8
        units = ((w + unit - 1) // unit) * ((h + unit - 1) // unit)
    markers = options.restart_interval is not None and options.restart_interval != 0 and options.restart_interval < units
    return not options.progressive and not markers


class _HostBand:
    """Stand-in for `jpeg.BandEncoder` on a band's tuple in host memory (tests: gloo on CPU; also the twin
    the device coder is checked against)."""

    def __init__(self, options, parts, index, coeff_fn):
        self.options = options
        b = jpeg.band(options.width, options.height, int(options.color_type), int(options.subsampling), parts, index)
        self.row_begin, self.row_end = b["row_begin"], b["row_end"]
        self.rows = self.row_end - self.row_begin
        self.coeff_fn = coeff_fn
        self.y = self.cb = self.cr = np.zeros((0, 64), np.int16)

    def close(self):
        pass

    def coeffs(self, band_pixels):
        if self.rows == 0:
            return [0, 0, 0]
        band_opts = jpeg.JpegOptions(**{**self.options.__dict__, "height": self.rows, "restart_interval": None})
        self.y, self.cb, self.cr = self.coeff_fn(band_pixels, band_opts)
        last = [int(self.y[-1, 0]), 0, 0]
        if self.cb.shape[0]:
            last[1], last[2] = int(self.cb[-1, 0]), int(self.cr[-1, 0])
        return last

    def count(self, prev_dc):
        if self.rows == 0:
            return np.zeros(jpeg.COUNT_WORDS, np.uint64)
        return jpeg.band_count_host(self.y, self.cb, self.cr, self.options, self.rows, prev_dc)

    def lengths(self, prev_dc, total_counts=None):
        if self.rows == 0:
            return 0
        return jpeg.band_bits_host(self.y, self.cb, self.cr, self.options, self.rows, prev_dc, total_counts)

    def pack(self, bit_offset):
        if self.rows == 0:
            return bytes(16)
        return jpeg.band_piece_host(self.y, self.cb, self.cr, self.options, self.rows, self._prev, bit_offset, self._counts)


def _all_gather_i64(values, group, device):
    """all_gather of a few integers per rank (the exchanges of the path are this small)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t = torch.tensor(values, dtype=torch.int64, device=device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return [o.cpu().tolist() for o in out]


_pinned = {}  # grow-only pinned staging for the finished file on the destination rank


def _pinned_file(n):
    import torch
    t = _pinned.get("file")
    if t is None or t.numel() < n:
        t = torch.empty(n + n // 4 + 4096, dtype=torch.uint8).pin_memory()
        _pinned["file"] = t
    return t


class SharedFile:
    """A file buffer in POSIX shared memory that every rank of ONE node maps (`SharedFile(name, size, create=rank == 0)`,
    the others after a barrier with `create=False`).  With it `encode_banded` lets every rank copy its band's body from its
    GPU over its OWN PCIe link straight to the body's final place in the file — no gather to one rank, whose xGMI links
    and single PCIe link would otherwise carry everybody's bytes.  `register()` pins this process's mapping
    (hipHostRegister) so that the copy is one DMA; without it the copy is staged through the library's pinned buffer."""

    def __init__(self, name, size, create):
        from multiprocessing import shared_memory
        self.shm = shared_memory.SharedMemory(name=name, create=create, size=size if create else 0)
        self.size = size
        self.registered = False

    def array(self):
        return np.ndarray((self.size,), dtype=np.uint8, buffer=self.shm.buf)

    def register(self):
        import torch
        if not self.registered:
            rc = torch.cuda.cudart().cudaHostRegister(self.array().ctypes.data, self.size, 0)
            self.registered = int(rc) == 0
        return self.registered

    def close(self, unlink=False):
        import torch
        if self.registered:
            torch.cuda.cudart().cudaHostUnregister(self.array().ctypes.data)
            self.registered = False
        self.shm.close()
        if unlink:
            self.shm.unlink()


def encode_banded(band_pixels, options, group=None, dst=0, device=None, coeff_fn=None, out=None, shared=None):
    """Collective over `group`.  Every rank passes the same `options` and ITS band's rows
    (`jpeg.band(w, h, ct, ss, world, rank)`: rows [row_begin, row_end), tightly packed) as host bytes /
    uint8 array or as a torch uint8 tensor on its GPU.  Returns the JFIF bytes on rank `dst`, None elsewhere;
    with `out` (a CPU uint8 tensor on `dst`, ideally pinned) the file is written there and its length returned.

    `shared`: a `SharedFile` every rank has mapped (one node): every rank writes its body into it over its own PCIe
    link; the file's length is returned on `dst` (the bytes are in `shared.array()`), None elsewhere.

    `device`: HIP device index of this rank (default: torch's current device); `coeff_fn(band_pixels,
    band_options) -> (y, cb, cr)`: tests substitute a CPU function, which also routes the entropy stage
    through the host twins, so that the whole exchange runs on gloo without a GPU."""
    import torch
    import torch.distributed as dist

    if not band_codable(options):
        raise ValueError("progressive scans and restart markers are not band-codable: use encode_gathered / "
                         "encode_gathered_device")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    on_gpu = coeff_fn is None
    prev_producer = None
    if on_gpu:
        if device is None:
            device = torch.cuda.current_device()
        enc = jpeg.BandEncoder(options, world, rank, device)
        tdev = torch.device("cuda", device)
        if hasattr(band_pixels, "is_cuda") and band_pixels.is_cuda:
            # (per thread and sticky: the previous setting is put back when the call returns, see `finally`)
            prev_producer = jpeg.get_producer_stream()
            jpeg.set_producer_stream(torch.cuda.current_stream(tdev).cuda_stream)
    else:
        enc = _HostBand(options, world, rank, coeff_fn)
        tdev = torch.device("cpu")
    try:
        rows = enc.row_end - enc.row_begin
        last = enc.coeffs(band_pixels)
        # exchange 1: boundary DCs (+ whether the band has rows)
        got = _all_gather_i64([rows > 0] + last, group, tdev)
        prev = [0, 0, 0]
        for r in range(rank):
            if got[r][0]:
                prev = got[r][1:4]
        total_counts = None
        if options.optimize_huffman:  # exchange 1b: symbol statistics, summed
            counts = enc.count(prev)
            t = torch.from_numpy(counts.view(np.int64).copy()).to(tdev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            total_counts = t.cpu().numpy().view(np.uint64)
        if not on_gpu:
            enc._prev, enc._counts = prev, total_counts
        bits = enc.lengths(prev, total_counts)
        # exchange 2: bits per band -> this band's bit offset
        all_bits = [b[0] for b in _all_gather_i64([bits], group, tdev)]
        offset = sum(all_bits[:rank])
        if not on_gpu and shared is not None:  # host twins through the shared file: header exchange, body at its final place
            piece = enc.pack(offset)
            words = np.frombuffer(piece[:16], np.int64)
            all_hdr = [np.array(h, np.int64).tobytes() for h in _all_gather_i64([int(words[0]), int(words[1])], group, tdev)]
            file_len, body_off = jpeg.splice_layout(options, all_hdr, total_counts)
            arr = shared.array()
            body = np.frombuffer(piece, np.uint8)[16:]
            arr[body_off[rank]: body_off[rank] + body.size] = body
            dist.barrier(group=group)
            if rank != dst:
                return None
            jpeg.splice_finish(options, all_hdr, arr, file_len, total_counts)
            return file_len
        if not on_gpu:  # host twins: whole pieces, gathered as objects
            piece = enc.pack(offset)
            pieces = [None] * world if rank == dst else None
            dist.gather_object(piece, pieces, dst=dst, group=group)
            if rank != dst:
                return None
            blob = jpeg.splice(options, pieces, total_counts)
            if out is None:
                return blob
            out[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
            return len(blob)
        # device: the body stays in HBM; exchange 3 = the 16-byte piece headers -> the file's layout
        hdr, n = enc.pack_device(offset)
        words = np.frombuffer(hdr, np.int64)
        all_hdr = [np.array(h, np.int64).tobytes() for h in _all_gather_i64([int(words[0]), int(words[1])], group, tdev)]
        file_len, body_off = jpeg.splice_layout(options, all_hdr, total_counts)
        lens = [int(np.frombuffer(h, np.uint64)[1]) for h in all_hdr]
        if shared is not None:
            # every rank: device -> the body's final place in the node's shared file, over this GPU's own PCIe link
            arr = shared.array()
            if lens[rank]:
                enc.copy_body(arr.ctypes.data + body_off[rank])
            dist.barrier(group=group)
            if rank != dst:
                return None
            jpeg.splice_finish(options, all_hdr, arr, file_len, total_counts)
            return file_len
        if world == 1:
            file = out if out is not None else _pinned_file(file_len)
            enc.copy_body(file[body_off[0]:])  # device -> its final place in the (pinned) file
        else:
            # the bodies travel to dst over xGMI (one gather of equal-sized buffers), then dst's PCIe link
            send = torch.empty(max(max(lens), 1), dtype=torch.uint8, device=tdev)
            enc.copy_body(send)
            recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
            dist.gather(send, recv, dst=dst, group=group)
            if rank != dst:
                return None
            file = out if out is not None else _pinned_file(file_len)
            for k in range(world):
                if lens[k]:
                    file[body_off[k]: body_off[k] + lens[k]].copy_(recv[k][: lens[k]], non_blocking=True)
            torch.cuda.synchronize(tdev)
        jpeg.splice_finish(options, all_hdr, file, file_len, total_counts)
        if out is not None:
            return file_len
        return file[:file_len].numpy().tobytes()
    finally:
        enc.close()
        if prev_producer is not None:
            jpeg.set_producer_stream(prev_producer)


def encode_gathered(data, options, group=None, coeff_fn=None, dst=0):
    """Fallback for option sets that are not band-codable: every rank computes its band's coefficients,
    the bands are gathered on `dst`, which runs the entropy stage over the stitched tuple (host arrays).
    `data`: the full image or at least this rank's rows at their place.  `trellis_quant` cannot be
    honoured here (the tuple entry points refuse it)."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    coeff_fn = coeff_fn or jpeg.coefficients
    w, h = options.width, options.height
    ct, ss = int(options.color_type), int(options.subsampling)
    bpp = 1 if ct == 0 else 3
    px = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1)
    b = jpeg.band(w, h, ct, ss, world, rank)
    rows = b["row_end"] - b["row_begin"]
    if rows > 0:
        if coeff_fn is jpeg.coefficients:
            import torch
            if torch.cuda.is_available():
                jpeg.set_device(torch.cuda.current_device())  # the host-pointer entries run on the thread's device
        sub = px[b["row_begin"] * w * bpp: b["row_end"] * w * bpp]
        band_opts = jpeg.JpegOptions(**{**options.__dict__, "height": rows, "restart_interval": None})
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


def encode_gathered_device(d_band_pixels, options, group=None, dst=0, coeff_fn=None, entropy_fn=None):
    """Device-resident form of `encode_gathered`: every rank holds ITS band as a torch uint8 tensor on its
    GPU; the finished coefficient bands travel to rank `dst` with one `torch.distributed.gather` per plane
    (RCCL over xGMI: peers send their share point to point), and `dst` runs the device entropy stage over
    the stitched tuple."""
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
        band_opts = jpeg.JpegOptions(**{**options.__dict__, "height": rows, "restart_interval": None})
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
    if fcb.shape[0] == 0:  # gray: the planes are unused but must be valid pointers
        fcb = fcr = fy
    with jpeg.producer_stream(torch.cuda.current_stream(dev).cuda_stream):  # the gathers above precede the entropy stage
        return jpeg.entropy_encode_device(fy, fcb, fcr, options)
