#!/bin/bash
# GPU call O: HBM traffic of the entropy kernels (FETCH_SIZE / WRITE_SIZE in separate passes, kernel trace only)
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2o; mkdir -p $O; export TMPDIR=/tmp
for c in ${COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $ROOT/tools/encode_loop.py 6 1 noise > /dev/null 2>&1)
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'].split('(')[0][-60:]].append(float(r['Counter_Value']))
for k, v in acc.items():
    print("%-12s %-62s launches %3d  mean %12.1f KiB (x2 for FETCH on gfx950)" % (sys.argv[2], k, len(v), sum(v) / len(v)))
PY
done 2>&1 | tee $O/pmc.txt
