#!/bin/bash
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2h; mkdir -p $O; export TMPDIR=/tmp
echo "== new"; timeout 120 python tools/probe_producer.py 2>&1 | grep -v amdgpu | tee $O/new.txt
echo "== old"; PIXO_HIP_OLD_ENTROPY=1 timeout 120 python tools/probe_producer.py 2>&1 | grep -v amdgpu | tee $O/old.txt
echo "== trace new"; (cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o e -- python $ROOT/tools/probe_producer.py > /dev/null 2>&1); f=$(find /tmp/pp -name "*kernel_stats*" | head -1); python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("  %-60s calls %4s avg %10.1f us max %10.1f" % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs'])/1e3, float(r['MaxNs'])/1e3))
PY
