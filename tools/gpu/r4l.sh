#!/bin/bash
# Round 3, GPU call 4l: length of the late start re-tuned after the gridDim load had gone (A/B).
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r4l; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do
  for v in new nost q45 q64 q90 t70 t100; do
    lib=""; [ $v != new ] && lib="$ROOT/pixo_amd/ab_$v.so"
    for wl in c2 c2_444; do
    PIXO_HIP_LIB=$lib timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl $v', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
    done
  done
done 2>&1 | tee $O/ab.txt
