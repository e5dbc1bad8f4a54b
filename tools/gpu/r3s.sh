#!/bin/bash
# Round 3, GPU call S: PNG kernel time per strategy (Bigrams re-measured), PNG suite.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3s; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== png per strategy"; timeout 300 python tools/png_probe.py 2>&1 | grep -v "$F" | tee $O/png_per_strategy.txt
echo "== long rows"; timeout 300 python tools/png_long_rows.py 2>&1 | grep -v "$F" | tail -12 | tee $O/png_long_rows.txt
ls $O
