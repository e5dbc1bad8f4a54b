#!/bin/bash
# Round 3, GPU call U: v_pack_b32_f16 for the quantiser's i16 pairs (A/B: parity first, then 4:4:4 / batch / c2 timings).
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3u; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== parity of the pack build"; PIXO_HIP_LIB=$ROOT/pixo_amd/ab_packf16.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not native_library and not fall_back" 2>&1 | grep -v "$F" | tail -4 | tee $O/pytest_pack.txt
for rep in 1 2; do
  for v in new packf16; do
    lib=""; [ $v != new ] && lib="$ROOT/pixo_amd/ab_$v.so"
    for wl in c2_444 c3 c2; do
      PIXO_HIP_LIB=$lib timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl $v', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
    done
  done
done 2>&1 | tee $O/ab_pack.txt
ls $O
