#!/bin/bash
# Round 3, GPU call 4f: batches in sub-batches over two contexts — parity + timings.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r4f; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest batch"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "batch" 2>&1 | grep -v "$F" | tail -4 | tee $O/pytest.txt
echo "== batch timings"; timeout 300 python tools/batch_restart_timing.py 2>&1 | grep -v "$F" | grep batch | tee $O/batch.txt
ls $O
