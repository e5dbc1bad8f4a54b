#!/bin/bash
# Round 3, GPU call C: the coefficient kernel with kernel arguments preloaded into SGPRs, cheap scalar addressing, 3-D grid and
# dot4 colour conversion: parity, A/B against the round-2 kernel on ONE box, the timeline again.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3c; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== parity (new default library)"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q 2>&1 | grep -v "$F" | tail -6 | tee $O/parity.txt
echo "== A/B"
ab() { # name args
  PIXO_HIP_LIB=$PWD/pixo_amd/ab_$1.so timeout 300 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-extras ${2:-} 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-10s %-24s kernel %7.2f us (blocks %6.2f .. %6.2f)  ms_per_step %.5f  frac %.3f' % ('$1', '${2:-c2}', r['kernel_us_avg'], r.get('kernel_us_block_min', 0), r.get('kernel_us_block_max', 0), d['ms_per_step'], r['frac']))"
}
{ for rep in 1 2 3; do for v in r02 new nopreload nodot4; do ab $v; done; done
  for wl in c3 c2_444 c2_unaligned; do for v in r02 new r02 new; do ab $v "--workload $wl"; done; done; } 2>&1 | tee $O/ab.txt
echo "== driver-form bench line of the new library"
PIXO_HIP_LIB=$PWD/pixo_amd/ab_new.so timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $O/bench_driver.json; cut -c1-600 $O/bench_driver.json
echo "== timeline"
PIXO_HIP_LIB=$PWD/pixo_amd/ab_probe.so timeout 200 python tools/probe_timeline.py c2 probe 2>&1 | grep -v "$F" > $O/timeline_c2.txt; cat $O/timeline_c2.txt
ls $O
