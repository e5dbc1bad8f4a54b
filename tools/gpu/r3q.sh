#!/bin/bash
# Round 3, GPU call Q: whole suite with direct unaligned loads and v_lerp_u8 as the shipped forms; c5 / unaligned timings.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3q; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -5 | tee $O/pytest.txt
for wl in c5 c2_unaligned c2 c2_444; do
  for rep in 1 2; do
    timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
  done
done 2>&1 | tee $O/timings.txt
rm -rf /tmp/pmc_x; (cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/pmc_x -o pmc -- python $ROOT/bench.py --workload c5 --steps 12 --warmup 4 --blocks 2 --settle-ms 0 --no-cpu-baseline --no-extras > /dev/null 2>&1)
f=$(find /tmp/pmc_x -name "*counter_collection*" | head -1); [ -n "$f" ] && python $ROOT/tools/pmc_summary.py "$f" png_filter | tee $O/pmc_c5.txt
ls $O
