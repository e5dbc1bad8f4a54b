#!/bin/bash
# Round 3, GPU call H: host pipeline v2 (upload thread + helper), result-block cache, fresh-page microbenchmark.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3h; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== fresh pages"; ./tools/ubench/bin/fresh_pages 178 2>&1 | tail -11 | tee $O/fresh_pages.txt
echo "== pytest (whole GPU suite)"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -12 | tee $O/pytest.txt
echo "== host pixel pipeline"
timeout 600 python tools/host_pipeline_timing.py 2>&1 | grep -v "$F" | tee $O/host_pipeline.txt
echo "== multi"
timeout 600 python tools/multi_timing.py 2>&1 | grep -v "$F" | tee $O/multi.txt
ls $O
