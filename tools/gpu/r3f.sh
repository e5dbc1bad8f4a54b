#!/bin/bash
# Round 3, GPU call F: batch in one copy (gaps for the headers between the scans), duplex-copy microbenchmark.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3f; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest (batch, fallback, restart)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "batch or fall_back or restart or goldens" 2>&1 | grep -v "$F" | tail -8 | tee $O/pytest.txt
echo "== batch / restart timings"
timeout 600 python tools/batch_restart_timing.py 2>&1 | grep -v "$F" | head -4 | tee $O/batch.txt
echo "== duplex"
./tools/ubench/bin/duplex 2>&1 | tee $O/duplex.txt
ls $O
