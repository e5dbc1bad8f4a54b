#!/bin/bash
# Round 3, GPU call A: coefficient-kernel variants (dot4 colour conversion, quantiser reciprocals in VGPRs via LDS):
# parity of each variant, A/B timing on ONE box, per-wavefront timeline of a dispatch (probe builds), split-launch probe.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3a; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
{ rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocm-smi --showclocks 2>/dev/null | head -20; } > $O/box.txt 2>&1
echo "== parity of the variants"
for v in dot4 ldsq both; do
  echo "--- $v"; PIXO_HIP_LIB=$PWD/pixo_amd/ab_$v.so timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "every_rgb or edge_dimensions or quality_sweep or saturated or config2 or goldens or batch_of_1080p or determinism" 2>&1 | grep -v "$F" | tail -4
done 2>&1 | tee $O/parity.txt
echo "== A/B"
ab() { # name args
  PIXO_HIP_LIB=$PWD/pixo_amd/ab_$1.so timeout 300 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-extras ${2:-} 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-6s %-24s value %9.0f Mpx/s  kernel %7.2f us (blocks %6.2f .. %6.2f)  ms_per_step %.5f  frac %.3f' % ('$1', '${2:-c2}', d['value'], r['kernel_us_avg'], r.get('kernel_us_block_min', 0), r.get('kernel_us_block_max', 0), d['ms_per_step'], r['frac']))"
}
{ for rep in 1 2 3; do for v in base dot4 ldsq both; do ab $v; done; done
  for wl in c3 c2_444 c2_unaligned; do for v in base both base both; do ab $v "--workload $wl"; done; done; } 2>&1 | tee $O/ab.txt
echo "== timelines"
for v in probe probe_both; do PIXO_HIP_LIB=$PWD/pixo_amd/ab_$v.so timeout 200 python tools/probe_timeline.py c2 $v 2>&1 | grep -v "$F" > $O/timeline_$v.txt; tail -8 $O/timeline_$v.txt; done
echo "== split launches"
PIXO_HIP_LIB=$PWD/pixo_amd/ab_base.so timeout 200 python tools/split_launch_probe.py 2>&1 | grep -v "$F" | tee $O/split.txt
ls $O
