#!/bin/bash
# Round 3, GPU call X: PNG kernel with v_perm byte extraction and AdaptiveFast as a template parameter.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3x; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest png"; timeout 900 python -m pytest tests/test_gpu_png.py -x -q 2>&1 | grep -v "$F" | tail -4 | tee $O/pytest_png.txt
echo "== png per strategy"; timeout 300 python tools/png_probe.py 2>&1 | grep -v "$F" | tee $O/png_per_strategy.txt
for rep in 1 2 3; do
  timeout 300 python3 bench.py --workload c5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c5', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
done 2>&1 | tee $O/c5.txt
timeout 300 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c2', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])" | tee $O/c2.txt
ls $O
