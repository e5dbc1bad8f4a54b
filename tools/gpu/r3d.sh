#!/bin/bash
# Round 3, GPU call D: segments (batches, restart intervals) in the single-pass entropy kernels, zero-copy batch delivery,
# bounded look-back with fallback: the whole GPU suite, then timings.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3d; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest (whole GPU suite)"
timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -v "$F" | tail -40 | tee $O/pytest.txt
echo "== batch / restart timings"
timeout 600 python tools/batch_restart_timing.py 2>&1 | grep -v "$F" | tee $O/batch_restart.txt
ls $O
