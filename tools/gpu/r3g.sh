#!/bin/bash
# Round 3, GPU call G: host-pixel pipeline with the helper thread, persistent band workers, result blocks: suite + timings.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3g; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>&1 | tee $O/thp.txt
echo "== pytest (whole GPU suite)"
timeout 1200 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | grep -v "$F" | tail -30 | tee $O/pytest.txt
echo "== host pixel pipeline"
timeout 600 python tools/host_pipeline_timing.py 2>&1 | grep -v "$F" | tee $O/host_pipeline.txt
echo "== multi"
timeout 600 python tools/multi_timing.py 2>&1 | grep -v "$F" | tee $O/multi.txt
ls $O
