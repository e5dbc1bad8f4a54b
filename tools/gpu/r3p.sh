#!/bin/bash
# Round 3, GPU call P: direct unaligned loads (A/B) + parity of that build on odd widths, stagger on/off, multi timing.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3p; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== parity of the direct-unaligned build"; PIXO_HIP_LIB=$ROOT/pixo_amd/ab_direct.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not fall_back and not pieces" 2>&1 | grep -v "$F" | tail -4 | tee $O/pytest_direct.txt
echo "== A/B unaligned workload"
for rep in 1 2 3; do
  for v in new direct; do
    lib=""; [ $v != new ] && lib="$ROOT/pixo_amd/ab_$v.so"
    PIXO_HIP_LIB=$lib timeout 300 python3 bench.py --workload c2_unaligned --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('unaligned $v', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
  done
done 2>&1 | tee $O/ab_unaligned.txt
echo "== A/B c2 stagger on/off"
for rep in 1 2 3; do
  for v in new nostagger; do
    lib=""; [ $v != new ] && lib="$ROOT/pixo_amd/ab_$v.so"
    PIXO_HIP_LIB=$lib timeout 300 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c2 $v', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
    PIXO_HIP_LIB=$lib timeout 300 python3 bench.py --workload c3 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c3 $v', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
  done
done 2>&1 | tee $O/ab_stagger.txt
echo "== multi"; timeout 300 python tools/multi_timing.py 2>&1 | grep -v "$F" | tee $O/multi.txt
ls $O
