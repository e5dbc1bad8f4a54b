#!/bin/bash
# Round 3, GPU call T: register-resident Bigrams kernel — PNG suite + per-strategy kernel times.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3t; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest png"; timeout 900 python -m pytest tests/test_gpu_png.py -x -q 2>&1 | grep -v "$F" | tail -6 | tee $O/pytest_png.txt
echo "== png per strategy"; timeout 300 python tools/png_probe.py 2>&1 | grep -v "$F" | tee $O/png_per_strategy.txt
ls $O
