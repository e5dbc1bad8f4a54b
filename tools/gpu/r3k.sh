#!/bin/bash
# Round 3, GPU call K: PNG filter kernel after the arithmetic diet (biased-lane Paeth, DPP wave sums, one checksum
# multiplication per group, scalar decision): parity, timing, instruction counts.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3k; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest png"; timeout 900 python -m pytest tests/test_gpu_png.py -x -q 2>&1 | grep -v "$F" | tail -8 | tee $O/pytest_png.txt
echo "== c5 timings"
for rep in 1 2 3; do
  timeout 300 python3 bench.py --workload c5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c5', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
done 2>&1 | tee $O/c5.txt
echo "== pmc"
for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $PMC | tr " " "_" | cut -c1-24)
  rm -rf /tmp/pmc_${tag}
  (cd /tmp && timeout 300 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmc_${tag} -o pmc -- python $ROOT/bench.py --workload c5 --steps 12 --warmup 4 --blocks 2 --settle-ms 0 --no-cpu-baseline --no-extras > /dev/null 2>&1)
  f=$(find /tmp/pmc_${tag} -name "*counter_collection*" | head -1)
  [ -n "$f" ] && { echo "--- c5 $PMC"; python $ROOT/tools/pmc_summary.py "$f" png_filter; }
done 2>&1 | tee $O/pmc_c5.txt
echo "== host path"
{ timeout 300 python tools/host_pipeline_timing.py; } 2>&1 | grep -v "$F" | tee $O/host_pipeline.txt
ls $O
