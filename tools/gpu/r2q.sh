#!/bin/bash
# GPU call Q: scans coded in pieces with overlapped delivery: parity (many small pieces, own size), whole-file timings
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2q; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q --durations=4 2>&1 | grep -v "$F" | tail -12 | tee $O/pytest.txt
for one in 0 1; do
  echo "== PIXO_HIP_ONE_PIECE=$one"
  PIXO_HIP_ONE_PIECE=$one timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_one$one.json; python -c "
import json; d=json.load(open('$O/bench_one$one.json')); print('ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'whole_file', d['whole_file']['ms_per_image'], d['whole_file']['ms_per_image_min'], d['whole_file']['ms_per_image_as_python_bytes'])"
  PIXO_HIP_ONE_PIECE=$one timeout 300 python3 bench.py --workload c4 --steps 5 --warmup 2 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c4 ms/step', d['ms_per_step'], d['value'])"
  PIXO_HIP_ONE_PIECE=$one PIXO_HIP_TRACE=1 timeout 100 python tools/e2e_device.py 2>&1 | grep -v "$F" | grep -v "reserve" | tail -14
done 2>&1 | tee $O/timing.txt
