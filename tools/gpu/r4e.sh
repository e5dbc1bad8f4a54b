#!/bin/bash
# Round 3, GPU call 4e: does the power-management performance level explain the spread between boxes?
# bench.py c2 / c2_444 / c5 at the default level (auto), at "high", and back at auto; clocks as rocm-smi reports them.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r4e; mkdir -p $O; export TMPDIR=/tmp
run() {
  for wl in c2 c2_444 c5; do
    timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $wl', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
  done
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Package Power" | sed "s/^/$1 /"
}
{
run auto
rocm-smi --setperflevel high 2>&1 | grep -v "^$" | head -5
run high
rocm-smi --setperfdeterminism 2400 2>&1 | grep -v "^$" | head -5
run determinism2400
rocm-smi --setperflevel auto 2>&1 | grep -v "^$" | head -3
run auto_again
} 2>&1 | tee $O/perf_level.txt
