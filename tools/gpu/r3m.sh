#!/bin/bash
# Round 3, GPU call M: staggered workgroup starts in the coefficient kernel (A/B builds), trellis sort on f64 min/max.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3m; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest progressive/trellis"; timeout 900 python -m pytest tests/test_gpu_progressive.py -x -q 2>&1 | grep -v "$F" | tail -5 | tee $O/pytest.txt
echo "== preset 2 timings"; timeout 300 python tools/preset2_timing.py 2>&1 | grep -v "$F" | tail -9 | tee $O/preset2.txt
echo "== A/B c2 stagger"
for rep in 1 2; do
  for v in new base st3x1 st8x1 st8x2 st9x1 st10x1 st10x2; do
    lib=""; [ $v != new ] && lib="$ROOT/pixo_amd/ab_$v.so"
    PIXO_HIP_LIB=$lib timeout 300 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
  done
done 2>&1 | tee $O/ab_c2.txt
echo "== trellis kernel stats"
rm -rf /tmp/prof_t; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python $ROOT/tools/preset2_timing.py > /dev/null 2>&1)
find /tmp/prof_t -name "*kernel_stats*" -exec cp {} $O/kernel_stats_preset2.csv \; ; cut -d, -f1-4 $O/kernel_stats_preset2.csv | cut -c1-150 | head -12
ls $O
