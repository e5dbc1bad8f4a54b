#!/bin/bash
# GPU call G: which tests got slow with the single-pass entropy kernels (durations), old entropy stage for comparison
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2g; mkdir -p $O; export TMPDIR=/tmp
echo "== new"; timeout 700 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -22 | tee $O/pytest_new.txt
echo "== old entropy"; PIXO_HIP_OLD_ENTROPY=1 timeout 300 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -10 | tee $O/pytest_old.txt
