#!/bin/bash
# Round 3, GPU call 4h: the grid shape as a preloaded argument (no scalar load of gridDim in front of every wavefront) — A/B.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r4h; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "coefficients or golden or batch" 2>&1 | grep -v "$F" | tail -3 | tee $O/pytest.txt
for rep in 1 2 3; do
  for v in new base; do
    lib=""; [ $v != new ] && lib="$ROOT/pixo_amd/ab_$v.so"
    for wl in c2 c2_444 c3; do
    PIXO_HIP_LIB=$lib timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl $v', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
    done
  done
done 2>&1 | tee $O/ab.txt
