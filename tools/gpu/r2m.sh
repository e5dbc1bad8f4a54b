#!/bin/bash
# GPU call M: stuffing kernel storing straight into pinned host memory: parity + whole-file timing, staged copy beside it
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2m; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest (entry points that deliver files)"; timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q --durations=3 2>&1 | grep -v "$F" | tail -8 | tee $O/pytest.txt
for staged in 0 1; do
  echo "== PIXO_HIP_DIRECT_STORES=$staged"
  PIXO_HIP_DIRECT_STORES=$staged timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_staged$staged.json; python -c "
import json; d=json.load(open('$O/bench_staged$staged.json')); print('ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'whole_file', d['whole_file'])"
  PIXO_HIP_DIRECT_STORES=$staged PIXO_HIP_TRACE=1 timeout 100 python tools/encode_loop.py 3 0 noise 2>&1 | grep -v "$F" | tail -12
done 2>&1 | tee $O/timing.txt
