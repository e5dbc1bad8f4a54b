#!/bin/bash
# GPU call S: trellis kernel with coalesced global loads instead of an LDS staging area: parity + preset-2 timings
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2s; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest"; timeout 600 python -m pytest tests/test_gpu_progressive.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -3 | tee $O/pytest.txt
timeout 200 python tools/preset2_timing.py 2>&1 | grep -v "$F" | tail -9 | tee $O/preset2.txt
rm -rf /tmp/prof_p2; (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p2 -o p2 -- python $ROOT/tools/preset2_timing.py > /dev/null 2>&1)
find /tmp/prof_p2 -name "*kernel_stats*" -exec cp {} $O/kernel_stats_preset2.csv \;
python - $O/kernel_stats_preset2.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("  %-60s calls %4s avg %9.1f us  %6s%%" % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
