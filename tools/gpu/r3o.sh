#!/bin/bash
# Round 3, GPU call O: whole GPU suite on the current tree, driver-form bench, preset-2 kernel times.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3o; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -5 | tee $O/pytest.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -2 | tee $O/smoke.txt
echo "== bench (driver form)"; timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $O/bench_driver.json; python3 -c "
import json
d=json.load(open('$O/bench_driver.json')); print(d['ms_per_step'], d['roofline']['kernel_us_avg'], d['roofline']['frac']); print({k:(v.get('kernel_us'),v.get('frac'),v.get('ms_per_image'),v.get('ms')) if isinstance(v,dict) else v for k,v in d.get('other_configs',{}).items()})"
echo "== preset 2"
rm -rf /tmp/prof_t; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python $ROOT/tools/preset2_timing.py 2>&1 | grep -v "$F" | grep Mpixels)
find /tmp/prof_t -name "*kernel_stats*" -exec cp {} $O/kernel_stats_preset2.csv \; ; python3 -c "
import csv
for r in csv.DictReader(open('$O/kernel_stats_preset2.csv')):
    if 'trellis' in r['Name'] or 'coeffs' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])"
echo "== no-stagger A/B"
ls $O
