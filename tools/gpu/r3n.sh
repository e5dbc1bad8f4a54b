#!/bin/bash
# Round 3, GPU call N: occupancy-limited and staggered variants of the coefficient kernel (A/B), trellis with key minima.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3n; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest progressive/trellis"; timeout 900 python -m pytest tests/test_gpu_progressive.py -x -q 2>&1 | grep -v "$F" | tail -3 | tee $O/pytest.txt
echo "== A/B c2"
for rep in 1 2; do
  for v in new occ4 occ5 occ6 st10u60 st10u90 st10u110 st10x1 st10x150 g9u40 g9u60; do
    lib=""; [ $v != new ] && lib="$ROOT/pixo_amd/ab_$v.so"
    PIXO_HIP_LIB=$lib timeout 300 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
  done
done 2>&1 | tee $O/ab_c2.txt
echo "== trellis kernel stats"
rm -rf /tmp/prof_t; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python $ROOT/tools/preset2_timing.py 2>&1 | grep -v "$F" | tail -9)
find /tmp/prof_t -name "*kernel_stats*" -exec cp {} $O/kernel_stats_preset2.csv \; ; grep trellis $O/kernel_stats_preset2.csv | cut -d, -f2-8
ls $O
