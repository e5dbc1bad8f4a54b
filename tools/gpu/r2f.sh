#!/bin/bash
# GPU call F: where the time of the code kernel goes (timing-only ablation builds of jpeg_scan_fused.hip; results are garbage on purpose)
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2f; mkdir -p $O; export TMPDIR=/tmp
for v in default fa1 fa2 fa3 fa4; do
  lib=""; [ $v != default ] && lib=$PWD/pixo_amd/ab_$v.so
  for k in "0 noise" "0 gradient"; do
    n=$(echo $k | tr " " "_")
    rm -rf /tmp/prof_${v}_$n
    (cd /tmp && PIXO_HIP_LIB=$lib timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${v}_$n -o e -- python $ROOT/tools/encode_loop.py 10 $k > /dev/null 2>&1)
    f=$(find /tmp/prof_${v}_$n -name "*kernel_stats*" | head -1)
    [ -n "$f" ] && python - "$f" "$v $k" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'scan_code' in r['Name'] or 'stuff_fused' in r['Name']:
        print("%-22s %-20s avg %8.1f us" % (sys.argv[2], r['Name'].split('(')[1][-24:] if False else ('code' if 'scan_code' in r['Name'] else 'stuff'), float(r['AverageNs'])/1e3))
PY
  done
done 2>&1 | tee $O/ablate.txt
