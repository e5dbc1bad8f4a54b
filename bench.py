#!/usr/bin/env python3
"""bench.py — Mpixels/s of the JPEG encode pixel pipeline (RGB→YCbCr→DCT→quant) on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic input that is already resident in HBM.

  --workload c2 (default, the metric; BASELINE.json configs[1]): one launch of the fused colour + DCT + quantise
      kernel over one 4096x4096 RGB8 image, q=80, 4:2:0.  Each rank rotates over enough distinct input/output
      buffers to exceed the 256 MiB Infinity Cache, so the bytes really come from and go to HBM.  N > 1: every rank
      encodes its own images (weak scaling, no data-path collective — SURVEY §8e).
  --workload c4 (configs[3]): ONE 16384x16384 image whose MCU-row bands live on the N GPUs; a step = the whole file:
      coefficient kernel + per-band entropy coding on every rank, the exchanges of pixo_amd/sharded.py (3 x i16 and a
      u64 per band over RCCL), the bodies gathered over xGMI, spliced on rank 0 (strong scaling).
  c2_444, c2_unaligned (4094 wide: rows not dword aligned), c3 (64 x 1080p, one launch), c1, c5 (PNG filters).

Timing: W warmup steps, then R blocks (--blocks) of EXACTLY K steps, each bracketed by a barrier and
torch.cuda.synchronize() on both sides, MAX over ranks per block; `ms_per_step` is the MEDIAN block (min and max
beside it).  Rank 0 prints ONE JSON line with the contract's keys plus `roofline`, `cpu_baseline`, `other_configs`.
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from benchlib.common import *  # noqa: E402,F401,F403
from benchlib.metric import run_coeffs, run_png  # noqa: E402
from benchlib.multi import run_c4, run_c4_single_process  # noqa: E402
from benchlib.roofline import issue_of, traffic_of  # noqa: E402,F401


def main():
    args = parse()
    job = Job(args)
    if args.workload == "c4" and args.single_process:
        return run_c4_single_process(job, args)
    if args.workload == "c5":
        return run_png(job, args)
    if args.workload == "c4":
        return run_c4(job, args)
    return run_coeffs(job, args)


if __name__ == "__main__":
    main()
