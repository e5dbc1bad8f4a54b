#!/bin/bash
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
for v in s1 s8 s13 s14 s20; do
  lib=$PWD/pixo_amd/ab_$v.so
  echo "== $v"; PIXO_HIP_LIB=$lib timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "goldens" 2>&1 | grep -v "$F" | tail -4
done
