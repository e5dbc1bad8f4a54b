#!/bin/bash
# GPU call E: fused entropy kernels after the wave-parallel look-back / 32-bit accumulator: parity + kernel times.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2e; mkdir -p $O; export TMPDIR=/tmp
echo "== (pytest skipped in this run)"
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
timeout 200 python tools/e2e_device.py 2>&1 | grep -v amdgpu | head -5 | tee -a $O/timing.txt
