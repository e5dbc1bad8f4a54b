#!/bin/bash
# Round 3, GPU call Y: late start for part of the PNG kernel's first generation (A/B builds).
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3y; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do
  for v in new p10x1 p10x2 p10x4 p9g1 p9g2 p8g1; do
    lib=""; [ $v != new ] && lib="$ROOT/pixo_amd/ab_$v.so"
    PIXO_HIP_LIB=$lib timeout 300 python3 bench.py --workload c5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c5 $v', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
  done
done 2>&1 | tee $O/ab_c5_stagger.txt
