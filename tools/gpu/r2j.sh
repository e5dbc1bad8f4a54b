#!/bin/bash
# GPU call J: single-walk code kernel: parity + kernel times
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2j; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest"; timeout 500 python -m pytest tests -m gpu -x -q --durations=3 2>&1 | grep -v "$F" | tail -8 | tee $O/pytest.txt
for k in "0 noise" "0 gradient" "1 noise"; do
  n=$(echo $k | tr " " "_")
  rm -rf /tmp/prof_e_$n
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e_$n -o e -- python $ROOT/tools/encode_loop.py 20 $k 2>&1 | grep "encode()")
  find /tmp/prof_e_$n -name "*kernel_stats*" -exec cp {} $O/kernel_stats_encode_$n.csv \;
  python - $O/kernel_stats_encode_$n.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("  %-70s calls %4s avg %9.1f us  min %8.1f max %8.1f" % (r['Name'].split('(')[0][-70:], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done 2>&1 | tee $O/timing.txt
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print('ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'whole_file', d['whole_file'])"
