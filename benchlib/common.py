"""bench.py's plumbing: constants, flags, the process group + device + timing protocol (Job), the ONE JSON line's way to stdout."""
import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):  # (the package pixo_amd; synth, oracle_lib — the checker and the CPU leg)
    if _p not in sys.path:
        sys.path.insert(0, _p)


HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


C4_SHA256 = "77cc6cb69a782693c46f2024ac57ebfdfb8411148fa3cef62698c727af36c70c"  # SURVEY §8c, made by the reference


WORKLOADS = {
    #               w      h   batch ss  label
    "c2": (4096, 4096, 1, 1, "configs[1]: single 4096x4096 RGB8, q=80, 4:2:0, fused colour+DCT+quant kernel"),
    "c2_444": (4096, 4096, 1, 0, "4096x4096 RGB8, q=80, 4:4:4"),
    "c2_unaligned": (4094, 4096, 1, 1, "4094x4096 RGB8 (rows not dword aligned: funnel loads), q=80, 4:2:0"),
    "c3": (1920, 1080, 64, 1, "configs[2]: batch of 64 x 1920x1080 RGB8, q=80, 4:2:0, one launch"),
    "c1": (512, 512, 1, 1, "configs[0] shape on the GPU: 512x512 RGB8, q=80, 4:2:0"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--blocks", type=int, default=15, help="R: the K-step block is timed R times; median/min/max are reported")
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="untimed launches before the warmup steps until this much wall time has passed: after an idle "
                         "period the GPU needs ~20 ms of work to reach its steady clocks (tools/warmup_probe.py); 0 = none")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS) + ["c4", "c5"])
    ap.add_argument("--quality", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip whole_file and other_configs (A/B runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    ap.add_argument("--single-process", action="store_true",
                    help="--workload c4 only: ONE process drives the N GPUs through pixo_hip_jpeg_encode_multi (a host thread per band, every "
                         "band over its own GPU's PCIe link) instead of one rank per GPU over RCCL; pixels start in HOST memory")
    ap.add_argument("--stub", action="store_true",
                    help="plumbing test without a GPU: gloo process group, the step is a short sleep (data: 'stub')")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# process plumbing: --gpus N is honoured whichever way the script is started
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def ensure_world(args):
    """Returns (rank, local_rank, world).  `python bench.py --gpus N` with N > 1 and no launcher environment
    re-executes itself as N ranks under torch.distributed.run; a launcher whose world size differs from --gpus is an error."""
    env_world = os.environ.get("WORLD_SIZE")
    if getattr(args, "single_process", False):
        if env_world not in (None, "1"):
            raise SystemExit("bench: --single-process is ONE process for all GPUs: do not start it under a multi-rank launcher")
        return 0, 0, 1
    if env_world is None and args.gpus > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + sys.argv[1:]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    claim_stdout()  # (not before the re-execution above: the ranks it starts inherit this process's descriptors)
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to report a number "
                         "for a different GPU count" % (args.gpus, world))
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world


class Job:
    """Process group + device + the timing protocol, shared by every workload."""

    def __init__(self, args):
        self.args = args
        self.rank, self.local_rank, self.world = ensure_world(args)
        self.stub = args.stub
        self.dist = None
        import torch
        self.torch = torch
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # TEST MODE (PIXO_BENCH_SHARE_GPU=1, tests/test_gpu_multi.py): the N ranks of a run all use GPU 0 and talk over gloo — RCCL
        # refuses two ranks on one device.  What a box with ONE GPU can check of an N-rank run: every leg's control flow and the
        # files' bytes at ranks above 0.  The line says so (`data`, `rccl.backend`); its numbers are not N-GPU numbers.
        self.share_gpu = bool(os.environ.get("PIXO_BENCH_SHARE_GPU")) and not self.stub and self.world > 1
        if self.stub:
            self.dev = self.wire = torch.device("cpu")
            self.gpu_index = None
            if self.dist is not None:
                self.dist.init_process_group(backend="gloo")
        else:
            self.gpu_index = 0 if self.share_gpu else self.local_rank
            torch.cuda.set_device(self.gpu_index)
            self.dev = torch.device("cuda", self.gpu_index)
            self.wire = torch.device("cpu") if self.share_gpu else self.dev  # where the tensors of the timing collectives live
            if self.dist is not None:
                if self.share_gpu:
                    self.dist.init_process_group(backend="gloo")
                else:
                    self.dist.init_process_group(backend="nccl", device_id=self.dev)

    def sync(self):
        if not self.stub:
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.sync()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.wire)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def settle(self, step, ms):
        """Untimed: keep the GPU busy for `ms` so that warmup and timed steps run at steady clocks."""
        n = 0
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < ms:
            for _ in range(16):
                step(n); n += 1
            self.sync()
        return n

    def time_blocks(self, step, steps, warmup, blocks, events=True):
        """W warmup steps, then `blocks` blocks of exactly `steps` steps: barrier + synchronize on both sides of every
        block, MAX over ranks.  Returns (wall seconds per block, HIP-event milliseconds per block on this rank)."""
        torch = self.torch
        for i in range(warmup):
            step(i)
        walls, evs = [], []
        n = warmup
        for _ in range(blocks):
            self.barrier()
            if events and not self.stub:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            t0 = time.perf_counter()
            for i in range(steps):
                step(n + i)
            if events and not self.stub:
                e1.record()
            self.sync()
            dt = time.perf_counter() - t0
            n += steps
            walls.append(self.max_over_ranks(dt))
            if events and not self.stub:
                evs.append(e0.elapsed_time(e1))
        self.barrier()
        return walls, evs

    def finish(self, line=None):
        """Tears the process group down, then (rank 0) prints the ONE JSON line — last, after whatever the
        runtime libraries still had in their stdio buffers (RCCL prints a version banner to stdout)."""
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
        if line is not None:
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            emit(line)


def block_stats(walls, steps):
    per = sorted(w / steps * 1e3 for w in walls)
    return {"ms_per_step": round(statistics.median(per), 5), "ms_per_step_min": round(per[0], 5), "ms_per_step_max": round(per[-1], 5),
            "blocks": len(per)}


_LINE_OUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner when its first
    communicator is made): from here on file descriptor 1 IS stderr, and only `emit` holds the real stdout."""
    global _LINE_OUT
    if _LINE_OUT is None:
        sys.stdout.flush()
        _LINE_OUT = os.fdopen(os.dup(1), "w", encoding="utf-8")
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    out = _LINE_OUT if _LINE_OUT is not None else sys.stdout
    out.write(json.dumps(line, ensure_ascii=False) + "\n")
    out.flush()


__all__ = [n for n in dir() if not n.startswith("__")]  # (everything, the private helpers and the stdlib modules included: the parts share one namespace)
