#!/bin/bash
# Round 3, GPU call 4i: PNG kernels with pre-loaded arguments (A/B against the struct-by-value form) + PNG suite.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r4i; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest png"; timeout 900 python -m pytest tests/test_gpu_png.py -x -q 2>&1 | grep -v "$F" | tail -3 | tee $O/pytest_png.txt
for rep in 1 2 3; do
  for v in new base; do
    lib=""; [ $v != new ] && lib="$ROOT/pixo_amd/ab_$v.so"
    PIXO_HIP_LIB=$lib timeout 300 python3 bench.py --workload c5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c5 $v', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
  done
done 2>&1 | tee $O/ab.txt
