#!/bin/bash
# GPU call N: symbol-statistics kernels (optimised tables): goldens + kernel times
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2n; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
for v in default; do
  if [ $v = default ]; then unset PIXO_HIP_LIB; else export PIXO_HIP_LIB=$PWD/pixo_amd/ab_$v.so; fi
  echo "== $v"; timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "goldens" 2>&1 | grep -v "$F" | tail -1
  for k in "1 noise" "1 gradient"; do
    n=$(echo $k | tr " " "_"); rm -rf /tmp/prof_e_$n
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e_$n -o e -- python $ROOT/tools/encode_loop.py 20 $k 2>&1 | grep "encode()")
    find /tmp/prof_e_$n -name "*kernel_stats*" -exec cp {} $O/kernel_stats_${v}_$n.csv \;
    python - $O/kernel_stats_${v}_$n.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'scan_c' in r['Name'] or 'stuff' in r['Name']:
        print("  %-60s calls %4s avg %9.1f us  min %8.1f max %8.1f" % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
  done
done 2>&1 | tee $O/timing.txt
