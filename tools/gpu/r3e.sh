#!/bin/bash
# Round 3, GPU call E: whole GPU suite again (fallback test fixed, host-pixel upload pipeline), batch / restart / host-pipeline timings.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3e; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest (whole GPU suite)"
timeout 1200 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | grep -v "$F" | tail -30 | tee $O/pytest.txt
echo "== batch / restart timings"
timeout 600 python tools/batch_restart_timing.py 2>&1 | grep -v "$F" | tee $O/batch_restart.txt
echo "== host pixel pipeline"
timeout 600 python tools/host_pipeline_timing.py 2>&1 | grep -v "$F" | tee $O/host_pipeline.txt
ls $O
