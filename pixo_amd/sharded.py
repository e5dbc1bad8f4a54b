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

A BATCH of images that is resident on ONE GPU (BASELINE config 3 on a node, SURVEY §8e "C3 batch") is the other
multi-GPU form, `encode_batch`: whole images travel from the source rank to the ranks point to point (one peer per
xGMI link, all posted at once), every rank encodes its images (src/jpeg/mod.rs:88 per image), and the finished FILES
travel to `dst` the same way — pixels out, files back, never a coefficient.
"""
import os

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


def _wire(group, tdev):
    """Where the tensors of a collective live.  RCCL takes device tensors as they are; gloo moves host memory only, so on a gloo
    group device tensors are staged through the host (nodes without RCCL between them — and the tests that run two or three
    ranks on ONE GPU, which RCCL refuses)."""
    import torch
    import torch.distributed as dist
    return torch.device("cpu") if dist.get_backend(group) == "gloo" else tdev


def _gather_bulk(t, dst, group, wire):
    """`dist.gather` of equally sized byte tensors; on `dst` the list of every rank's tensor (host tensors when staged)."""
    import torch
    import torch.distributed as dist
    send = t if t.device == wire or wire.type != "cpu" else t.cpu()
    recv = [torch.empty_like(send) for _ in range(dist.get_world_size(group))] if dist.get_rank(group) == dst else None
    dist.gather(send, recv, dst=_global(group, dst), group=group)
    return recv


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
    (hipHostRegister) so that the copy is one DMA; without it the copy is staged through the library's pinned buffer.

    Who made a segment is written INTO it (a 4 KiB header page in front of the file's bytes): magic, the maker's pid,
    its start time (clock ticks since boot, /proc/<pid>/stat — a reused pid has another) and its PID namespace.  A segment of the
    same name that exists already is replaced only when its maker is provably gone: same namespace as ours and no live process
    with that pid AND start time.  A maker in another PID namespace (containers sharing /dev/shm) cannot be judged from here: its
    segment is left alone.  A header that is still all zero belongs to a maker between shm_open and its first store: waited for
    briefly.  No companion segment, nothing for the resource tracker to complain about."""

    _HEADER = 4096
    _MAGIC = b"PIXOSHM2"

    def __init__(self, name, size, create):
        from multiprocessing import shared_memory
        import time
        total = size + self._HEADER
        try:
            self.shm = shared_memory.SharedMemory(name=name, create=create, size=total if create else 0)
        except FileExistsError:
            old = shared_memory.SharedMemory(name=name, create=False)
            self._untrack(old)
            owner = None
            for _ in range(50):  # (a maker that has not written its header yet: up to half a second)
                owner = self._read_header(old.buf)
                if owner is not None:
                    break
                time.sleep(0.01)
            mine = owner is not None and owner[0] == os.getpid() and owner[1] == self._start_time(os.getpid())
            if owner is not None and not mine and self._maybe_alive(owner):
                old.close()
                raise FileExistsError("SharedFile %r is in use by live process %d: names must be unique per job" % (name, owner[0]))
            old.close()
            try:
                shared_memory.SharedMemory(name=name, create=False).unlink()  # (stale — its maker is gone — or this process's own: replaced)
            except FileNotFoundError:
                pass
            self.shm = shared_memory.SharedMemory(name=name, create=True, size=total)
        if create:
            self._write_header(self.shm.buf)
        else:
            self._untrack(self.shm)
            if self.shm.size < total:
                n = self.shm.size
                self.shm.close()
                raise ValueError("SharedFile %r holds %d bytes, %d wanted" % (name, max(0, n - self._HEADER), size))
        self.size = size
        self.registered = False

    @staticmethod
    def _untrack(shm):
        # (Python < 3.13 registers attached segments with the resource tracker too, which then unlinks — or complains about — a
        # segment this process does not own when the process ends)
        try:
            from multiprocessing import resource_tracker
            resource_tracker.unregister(shm._name, "shared_memory")
        except Exception:
            pass

    @staticmethod
    def _start_time(pid):
        try:
            with open("/proc/%d/stat" % pid) as fh:
                return int(fh.read().rsplit(")", 1)[1].split()[19])  # field 22: starttime
        except Exception:
            return 0

    @staticmethod
    def _pid_namespace():
        try:
            return os.stat("/proc/self/ns/pid").st_ino
        except Exception:
            return 0

    def _write_header(self, buf):
        pid = os.getpid()
        buf[8:32] = pid.to_bytes(8, "little") + self._start_time(pid).to_bytes(8, "little") + self._pid_namespace().to_bytes(8, "little")
        buf[:8] = self._MAGIC  # (last: a reader that sees the magic sees the fields)

    @classmethod
    def _read_header(cls, buf):
        if len(buf) < 32 or bytes(buf[:8]) != cls._MAGIC:
            return None if len(buf) >= 32 and bytes(buf[:32]) == bytes(32) else (0, 0, 0)  # all zero: being made; anything else: not ours to judge -> stale
        return tuple(int.from_bytes(bytes(buf[8 + 8 * i: 16 + 8 * i]), "little") for i in range(3))

    @classmethod
    def _maybe_alive(cls, owner):
        pid, start, ns = owner
        if pid == 0:
            return False  # (a segment without a header: made by an older version, or garbage)
        if ns != cls._pid_namespace():
            return True   # (another PID namespace: cannot be judged from here — never unlink somebody else's segment)
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        except PermissionError:
            pass
        now = cls._start_time(pid)
        return now == 0 or start == 0 or now == start  # (a reused pid has another start time)

    def array(self):
        return np.ndarray((self.size,), dtype=np.uint8, buffer=self.shm.buf, offset=self._HEADER)

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
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass


def shared_file_bound(options) -> int:
    """A `SharedFile` size NO baseline file of these options can exceed: 2 KiB of headers + 416 bytes per 8x8 block (a DC
    symbol of <= 16 + 11 bits and 63 AC symbols of <= 16 + 10 bits are 1,665 bits = 209 bytes, and byte stuffing at most
    doubles them).  Real files are far smaller (noise at quality 80: 11 bytes per block) and untouched pages of a shared
    segment cost nothing, but a segment that is `register()`ed is pinned whole: choose a smaller one from what the content
    is known to need — `encode_banded` raises BufferTooSmall (`.needed`) on EVERY rank, before any byte moves, when the
    file turns out larger than the segment."""
    w, h = options.width, options.height
    blocks = ((w + 7) // 8) * ((h + 7) // 8) * (1 if int(options.color_type) == 0 else 3)
    return 2048 + 416 * blocks


def _check_shared(shared, file_len):
    """All ranks computed the same `file_len` from the gathered piece headers: all raise, nobody is left in a barrier."""
    if file_len > shared.size:
        from . import error
        e = error.BufferTooSmall("output buffer too small: the file needs %d bytes, the shared segment has %d"
                                 % (file_len, shared.size))
        e.needed = file_len
        raise e


def encode_banded(band_pixels, options, group=None, dst=0, device=None, coeff_fn=None, out=None, shared=None, phases=None):
    """Collective over `group`.  Every rank passes the same `options` and ITS band's rows
    (`jpeg.band(w, h, ct, ss, world, rank)`: rows [row_begin, row_end), tightly packed) as host bytes /
    uint8 array or as a torch uint8 tensor on its GPU.  Returns the JFIF bytes on rank `dst`, None elsewhere;
    with `out` (a CPU uint8 tensor on `dst`, ideally pinned) the file is written there and its length returned.

    `shared`: a `SharedFile` every rank has mapped (one node): every rank writes its body into it over its own PCIe
    link; the file's length is returned on `dst` (the bytes are in `shared.array()`), None elsewhere.  The segment's size
    has to be chosen before the file's length is known: `shared_file_bound(options)` is always enough (and usually far too much); a segment that
    turns out too small makes EVERY rank raise `BufferTooSmall` (`.needed` = the file's length) before any byte is written.

    `phases`: a dict that receives this rank's wall milliseconds per step (coeffs_ms, exchange_dc_ms, lengths_ms, exchange_bits_ms,
    pack_ms, exchange_hdr_ms, bodies_ms, splice_ms, total_ms): every step of the band encoder waits for its kernels itself, so the
    laps need no extra synchronisation.

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
    wire = _wire(group, tdev)
    import time
    t_start = t_lap = time.perf_counter()

    def lap(name):
        nonlocal t_lap
        if phases is not None:
            t1 = time.perf_counter()
            phases[name] = phases.get(name, 0.0) + (t1 - t_lap) * 1e3
            phases["total_ms"] = (t1 - t_start) * 1e3
            t_lap = t1
    try:
        rows = enc.row_end - enc.row_begin
        last = enc.coeffs(band_pixels)
        lap("coeffs_ms")
        # exchange 1: boundary DCs (+ whether the band has rows)
        got = _all_gather_i64([rows > 0] + last, group, wire)
        lap("exchange_dc_ms")
        prev = [0, 0, 0]
        for r in range(rank):
            if got[r][0]:
                prev = got[r][1:4]
        total_counts = None
        if options.optimize_huffman:  # exchange 1b: symbol statistics, summed
            counts = enc.count(prev)
            t = torch.from_numpy(counts.view(np.int64).copy()).to(wire)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            total_counts = t.cpu().numpy().view(np.uint64)
        if not on_gpu:
            enc._prev, enc._counts = prev, total_counts
        bits = enc.lengths(prev, total_counts)
        lap("lengths_ms")
        # exchange 2: bits per band -> this band's bit offset
        all_bits = [b[0] for b in _all_gather_i64([bits], group, wire)]
        offset = sum(all_bits[:rank])
        lap("exchange_bits_ms")
        if not on_gpu and shared is not None:  # host twins through the shared file: header exchange, body at its final place
            piece = enc.pack(offset)
            words = np.frombuffer(piece[:16], np.int64)
            all_hdr = [np.array(h, np.int64).tobytes() for h in _all_gather_i64([int(words[0]), int(words[1])], group, wire)]
            file_len, body_off = jpeg.splice_layout(options, all_hdr, total_counts)
            _check_shared(shared, file_len)
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
        lap("pack_ms")
        words = np.frombuffer(hdr, np.int64)
        all_hdr = [np.array(h, np.int64).tobytes() for h in _all_gather_i64([int(words[0]), int(words[1])], group, wire)]
        file_len, body_off = jpeg.splice_layout(options, all_hdr, total_counts)
        lens = [int(np.frombuffer(h, np.uint64)[1]) for h in all_hdr]
        lap("exchange_hdr_ms")
        if shared is not None:
            # every rank: device -> the body's final place in the node's shared file, over this GPU's own PCIe link
            _check_shared(shared, file_len)
            arr = shared.array()
            if lens[rank]:
                enc.copy_body(arr.ctypes.data + body_off[rank])
            dist.barrier(group=group)
            lap("bodies_ms")
            if rank != dst:
                return None
            jpeg.splice_finish(options, all_hdr, arr, file_len, total_counts)
            lap("splice_ms")
            return file_len
        if world == 1:
            file = out if out is not None else _pinned_file(file_len)
            enc.copy_body(file[body_off[0]:])  # device -> its final place in the (pinned) file
        else:
            # the bodies travel to dst over xGMI (one gather of equal-sized buffers), then dst's PCIe link
            send = torch.empty(max(max(lens), 1), dtype=torch.uint8, device=tdev)
            enc.copy_body(send)
            recv = _gather_bulk(send, dst, group, wire)
            if rank != dst:
                lap("bodies_ms")
                return None
            file = out if out is not None else _pinned_file(file_len)
            for k in range(world):
                if lens[k]:
                    file[body_off[k]: body_off[k] + lens[k]].copy_(recv[k][: lens[k]], non_blocking=True)
            torch.cuda.synchronize(tdev)
        lap("bodies_ms")
        jpeg.splice_finish(options, all_hdr, file, file_len, total_counts)
        lap("splice_ms")
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
        parts = _gather_bulk(tb, dst, group, _wire(group, dev))
        outs.append([p.to(dev).view(torch.int16) for p in parts] if parts is not None else None)
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


# ----------------------------------------------------------------------------------------------------------------------
# a batch of images resident on one GPU, encoded by all of them (SURVEY §8e "C3 batch")
# ----------------------------------------------------------------------------------------------------------------------
def batch_partition(n, world):
    """Images [lo, hi) of rank r: contiguous runs whose sizes differ by at most one, the longer ones first (64 images over
    8 ranks: 8 each; 5 images over 8 ranks: ranks 0-4 one each, ranks 5-7 none)."""
    base, extra = divmod(n, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def _global(group, r):
    import torch.distributed as dist
    return dist.get_global_rank(group, r) if group is not None else r


def _p2p_post(ops, group=None):
    """Posts all sends / receives of one step at once (RCCL runs them as one group: the source's seven xGMI links carry
    seven different peers' images at the same time) WITHOUT waiting.  ops: ("send" | "recv", tensor, global peer rank).
    On a gloo group device tensors travel through host copies (`_wire`).  Returns a handle for `_p2p_wait`."""
    import torch
    import torch.distributed as dist
    if not ops:
        return None
    staged_wire = dist.get_backend(group) == "gloo"
    posted, landed = [], []
    for kind, t, peer in ops:
        w = t
        if staged_wire and t.is_cuda:
            if kind == "send":
                w = t.cpu()
            else:
                w = torch.empty(t.shape, dtype=t.dtype)
                landed.append((t, w))
        posted.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, w, peer, group))
    return dist.batch_isend_irecv(posted), landed, posted


def _p2p_wait(handle):
    if handle is None:
        return
    reqs, landed, _ = handle
    for req in reqs:
        req.wait()
    for t, w in landed:
        t.copy_(w)


def _p2p(ops, group=None):
    """post + wait"""
    _p2p_wait(_p2p_post(ops, group))


def encode_batch(batch_pixels, options, n, group=None, src=0, dst=0, device=None, encode_fn=None, out=None, shared=None, waves=1, phases=None):
    """Collective over `group`: `n` equally sized images (described by `options`) lie back to back on rank `src` — a torch
    uint8 tensor on its GPU (other ranks pass None) — and come back as `n` JFIF files on rank `dst`, each byte-identical to
    `pixo::jpeg::encode` of that image (src/jpeg/mod.rs:88).

      1. scatter   whole images, `batch_partition(n, world)`, point to point from `src` (send/recv of 6.2 MB images for
                   1080p, one peer per link; `src` keeps its own share where it is)
      2. encode    every rank: `pixo_hip_jpeg_encode_batch_device_into` into an arena in ITS HBM (one coefficient launch,
                   one pass of the device entropy stage, files complete with headers)
      3. sizes     one all_gather of the per-image file lengths (8 bytes per image)
      4. gather    every rank sends its run of files to `dst`, which receives each run at its final offset of ONE device
                   arena and copies that arena to the host once

    Returns `(arena, offsets, lens)` on `dst` — `arena` a CPU uint8 tensor (`out` if given: ideally pinned; BufferTooSmall
    with `.needed` when it is too small), file i = `arena[offsets[i]: offsets[i] + lens[i]]` — and None elsewhere.

    `shared`: a `SharedFile` every rank of the node has mapped — step 4 then is: every rank copies ITS run of files from its
    GPU over its OWN PCIe link to the run's final offset in the shared arena (one barrier; nothing travels over xGMI and
    rank `dst`'s single link does not carry everybody's bytes: 89 MB for 64 x 1080p noise).  Returns `(None, offsets, lens)`
    on `dst` (the bytes are in `shared.array()`); a segment that is too small makes EVERY rank raise BufferTooSmall.

    `waves` (1 or 2): every share travels and is encoded in that many parts — the second part's images are on the wire while the
    first part's are encoded (the scatter of 348 MB of pixels is the long step of 64 x 1080p on a node: 7 links x ~50 GB/s);
    in either case `src` encodes its own share while its sends are in flight.  The files are the same bytes.

    `phases`: a dict that receives this rank's wall milliseconds per step (scatter_wait_ms, encode_ms, sizes_ms, gather_ms or
    copy_ms, total_ms) — the device is synchronised at the step boundaries then, so a call with `phases` is for diagnosis,
    not for timing the whole.

    A rank whose encode step fails (or, with `shared`, a segment that is too small) makes EVERY rank raise before any file
    moves: the size exchange carries a status word.

    `encode_fn(images, options, count) -> list of bytes` replaces step 2 for tests without a GPU (gloo, CPU tensors: the
    scatter, the size exchange and the gather are the same calls)."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    on_gpu = encode_fn is None
    bpp = 1 if int(options.color_type) == 0 else 3
    px = options.width * options.height * bpp
    parts = batch_partition(n, world)
    lo, hi = parts[rank]
    cnt = hi - lo
    if on_gpu:
        if device is None:
            device = torch.cuda.current_device()
        tdev = torch.device("cuda", device)
    else:
        tdev = torch.device("cpu")
    gsrc, gdst = _global(group, src), _global(group, dst)

    import time
    t_start = time.perf_counter()

    def lap(name, t0):
        if phases is None:
            return time.perf_counter()
        if on_gpu:
            torch.cuda.synchronize(tdev)
        t1 = time.perf_counter()
        phases[name] = phases.get(name, 0.0) + (t1 - t0) * 1e3
        return t1

    if world == 1 and on_gpu:
        # nothing to scatter or gather: the files go straight from this GPU into the destination's host arena — the caller's,
        # or the node-shared segment (registered: the device-to-host copies land in it directly) — while the library overlaps
        # their way over PCIe with the kernels of the next sub-batch: no device arena, no second pass over the bytes
        if batch_pixels is None or batch_pixels.numel() != n * px:
            raise ValueError("encode_batch: rank src passes the %d images back to back (%d bytes)" % (n, n * px))
        prev = jpeg.get_producer_stream()
        jpeg.set_producer_stream(torch.cuda.current_stream(tdev).cuda_stream)
        try:
            if shared is not None:
                arena = torch.from_numpy(shared.array())
            else:
                arena = out if out is not None else _pinned_file(n * (px // 2 + 4096))
            while True:
                try:
                    offsets, lens = jpeg.encode_batch_device_into(arena, batch_pixels.reshape(-1), options, n)
                    lap("encode_ms", t_start)
                    if phases is not None:
                        phases["total_ms"] = (time.perf_counter() - t_start) * 1e3
                    return (None if shared is not None else arena), [int(x) for x in offsets], [int(x) for x in lens]
                except jpeg.error.BufferTooSmall as e:
                    if out is not None or shared is not None:
                        raise
                    arena = _pinned_file(int(e.needed))
        finally:
            jpeg.set_producer_stream(prev)

    # 1. scatter — posted, not waited for: `src` goes on to its own share while its sends are in flight; with two waves a
    # receiver encodes the first part of its share while the second part travels
    nw = 2 if (waves == 2 and n >= 2 * world) else 1

    def cut(a, b):  # the parts of a share, in image indices
        if nw == 1 or b - a < 2:
            return [(a, b)]
        m = (a + b + 1) // 2
        return [(a, m), (m, b)]
    my_parts = cut(lo, hi)
    pending = []
    if rank == src:
        if batch_pixels is None or batch_pixels.numel() != n * px:
            raise ValueError("encode_batch: rank src passes the %d images back to back (%d bytes)" % (n, n * px))
        whole = batch_pixels.reshape(-1)
        mine = whole[lo * px: hi * px]
        for w in range(nw):
            ops = []
            for r, (a, b) in enumerate(parts):
                pr = cut(a, b)
                if r != src and w < len(pr) and pr[w][1] > pr[w][0]:
                    ops.append(("send", whole[pr[w][0] * px: pr[w][1] * px], _global(group, r)))
            pending.append(_p2p_post(ops, group))
        arrivals = [None] * len(my_parts)  # (its own images are where they are)
    else:
        mine = torch.empty(cnt * px, dtype=torch.uint8, device=tdev)
        arrivals = [_p2p_post([("recv", mine[(a - lo) * px: (b - lo) * px], gsrc)], group) if b > a else None for (a, b) in my_parts]
    t_lap = lap("scatter_post_ms", t_start)

    # 2. encode this rank's images part by part; the files stay, back to back, where the next step sends them from.  A
    # failure here must not leave the other ranks waiting in step 3: it travels as a status word.
    lens, failure = [], None
    run = torch.empty(0, dtype=torch.uint8, device=tdev)
    try:
        if not on_gpu:
            files = []
            for k, (a, b) in enumerate(my_parts):
                _p2p_wait(arrivals[k])
                t_lap = lap("scatter_wait_ms", t_lap)
                if b > a:
                    files += encode_fn(mine[(a - lo) * px: (b - lo) * px].numpy(), options, b - a)
                t_lap = lap("encode_ms", t_lap)
            lens = [len(f) for f in files]
            if cnt and sum(lens):
                run = torch.frombuffer(bytearray(b"".join(files)), dtype=torch.uint8)
        elif cnt:
            cap = cnt * (px // 2 + 4096)
            prev = jpeg.get_producer_stream()
            jpeg.set_producer_stream(torch.cuda.current_stream(tdev).cuda_stream)  # the received pixels were written on torch's stream
            try:
                run = torch.empty(cap, dtype=torch.uint8, device=tdev)
                at_run = 0
                for k, (a, b) in enumerate(my_parts):
                    _p2p_wait(arrivals[k])
                    t_lap = lap("scatter_wait_ms", t_lap)
                    if b <= a:
                        continue
                    while True:
                        try:
                            _, part_lens = jpeg.encode_batch_device_into(run[at_run:], mine[(a - lo) * px: (b - lo) * px], options, b - a)
                            break
                        except jpeg.error.BufferTooSmall as e:  # (the lengths were filled in: the second attempt fits)
                            grown = torch.empty(at_run + int(e.needed) + (hi - b) * (px // 2 + 4096), dtype=torch.uint8, device=tdev)
                            grown[:at_run].copy_(run[:at_run])
                            run = grown
                    lens += [int(x) for x in part_lens]
                    at_run += sum(int(x) for x in part_lens)
                    t_lap = lap("encode_ms", t_lap)
            finally:
                jpeg.set_producer_stream(prev)
    except Exception as ex:  # noqa: BLE001 — reported to every rank below
        failure = ex
        lens = [0] * cnt
    for h in pending:  # (src: its sends have long left; they must have before the tensors they read are released)
        _p2p_wait(h)
    t_lap = lap("scatter_wait_ms", t_lap)

    # 3. sizes: every rank's per-image lengths, padded to the longest share
    most = max(b - a for a, b in parts)
    my_cap = int(out.numel()) if (out is not None and rank == dst and shared is None) else -1  # (dst's fixed output, if any)
    gathered = _all_gather_i64([0 if failure is None else -1, my_cap] + lens + [0] * (most - cnt), group, _wire(group, tdev))
    bad = [r for r in range(world) if gathered[r][0] != 0]
    if bad:  # every rank knows: nobody posts a send or a receive, everybody raises
        if failure is not None:
            raise failure
        raise RuntimeError("encode_batch: the encode step failed on rank(s) %s of the group" % bad)
    all_lens = [g[2:] for g in gathered]
    dst_cap = gathered[dst][1]
    t_lap = lap("sizes_ms", t_lap)
    lens_all = [x for r, (a, b) in enumerate(parts) for x in all_lens[r][: b - a]]
    offsets, at = [], 0
    for x in lens_all:
        offsets.append(at)
        at += x
    run_len = [sum(all_lens[r][: b - a]) for r, (a, b) in enumerate(parts)]
    run_off = [offsets[a] if b > a else at for (a, b) in parts]

    # 4. (shared arena) every rank writes its own run at its final offset, over its own link
    if shared is not None:
        _check_shared(shared, at)  # (every rank computed the same total: all raise, nobody is left in the barrier)
        if run_len[rank]:
            dest = torch.from_numpy(shared.array()[run_off[rank]: run_off[rank] + run_len[rank]])
            dest.copy_(run[: run_len[rank]])  # (device -> the mapping: a DMA when the segment is registered, staged otherwise)
            if on_gpu:
                torch.cuda.synchronize(tdev)
        dist.barrier(group=group)
        lap("copy_ms", t_lap)
        if phases is not None:
            phases["total_ms"] = (time.perf_counter() - t_start) * 1e3
        return (None, offsets, lens_all) if rank == dst else None
    # 4. gather the files on dst, every run straight to its final offset.  dst's output too small: every rank knows (its
    # capacity travelled with the sizes) and raises the same error before a byte moves.
    if 0 <= dst_cap < at:
        from . import error
        e = error.BufferTooSmall("output buffer too small: need %d bytes" % at)
        e.needed = at
        raise e
    if rank != dst:
        if run_len[rank]:
            _p2p([("send", run[: run_len[rank]], gdst)], group)
        lap("gather_ms", t_lap)
        if phases is not None:
            phases["total_ms"] = (time.perf_counter() - t_start) * 1e3
        return None
    whole = torch.empty(max(at, 1), dtype=torch.uint8, device=tdev)
    if run_len[rank]:
        whole[run_off[rank]: run_off[rank] + run_len[rank]].copy_(run[: run_len[rank]])
    _p2p([("recv", whole[run_off[r]: run_off[r] + run_len[r]], _global(group, r)) for r in range(world) if r != dst and run_len[r]], group)
    if not on_gpu:
        arena = whole if out is None else out
        if out is not None:
            out[:at].copy_(whole[:at])
        lap("gather_ms", t_lap)
        if phases is not None:
            phases["total_ms"] = (time.perf_counter() - t_start) * 1e3
        return arena, offsets, lens_all
    t_lap = lap("gather_ms", t_lap)
    arena = out if out is not None else _pinned_file(at)
    arena[:at].copy_(whole[:at], non_blocking=True)
    torch.cuda.synchronize(tdev)
    lap("copy_ms", t_lap)
    if phases is not None:
        phases["total_ms"] = (time.perf_counter() - t_start) * 1e3
    return arena, offsets, lens_all
