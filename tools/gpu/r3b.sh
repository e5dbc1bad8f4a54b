#!/bin/bash
# Round 3, GPU call B: the dot4 variant's parity failure in detail, the back-to-back timeline (probe v2), where kernel
# arguments live (HIP_FORCE_DEV_KERNARG), and the whole GPU suite on the library after the capi.cpp split.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3b; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== dot4 parity (all goldens, no -x)"
PIXO_HIP_LIB=$PWD/pixo_amd/ab_dot4.so timeout 400 python -m pytest tests/test_gpu_parity.py -q --tb=short -k "goldens or every_rgb or edge_dimensions or saturated" 2>&1 | grep -v "$F" | tail -60 > $O/dot4_parity.txt; tail -45 $O/dot4_parity.txt
echo "== timeline"
PIXO_HIP_LIB=$PWD/pixo_amd/ab_probe.so timeout 200 python tools/probe_timeline.py c2 probe 2>&1 | grep -v "$F" > $O/timeline_c2.txt; cat $O/timeline_c2.txt
echo "== kernarg placement"
ab() { # label env
  env $2 PIXO_HIP_LIB=$PWD/pixo_amd/ab_base.so timeout 300 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-28s kernel %7.2f us (blocks %6.2f .. %6.2f)  ms_per_step %.5f  frac %.3f' % ('$1', r['kernel_us_avg'], r.get('kernel_us_block_min', 0), r.get('kernel_us_block_max', 0), d['ms_per_step'], r['frac']))"
}
{ for rep in 1 2; do ab default "A=1"; ab HIP_FORCE_DEV_KERNARG=1 "HIP_FORCE_DEV_KERNARG=1"; ab HIP_FORCE_DEV_KERNARG=0 "HIP_FORCE_DEV_KERNARG=0"; done; } 2>&1 | tee $O/kernarg.txt
echo "== pytest (whole GPU suite, default library)"
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -v "$F" | tail -16 | tee $O/pytest.txt
ls $O
