#!/bin/bash
# Round 3, GPU call L: the quantiser on packed multiply-adds — parity suite, A/B against the previous build (ab_base.so),
# instruction counts.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3l; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest (parity + progressive)"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_progressive.py -x -q 2>&1 | grep -v "$F" | tail -6 | tee $O/pytest.txt
echo "== A/B c2"
for rep in 1 2 3; do
  for v in base new; do
    lib=""; [ $v = base ] && lib="$ROOT/pixo_amd/ab_base.so"
    PIXO_HIP_LIB=$lib timeout 300 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
  done
done 2>&1 | tee $O/ab_c2.txt
echo "== driver form"
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_driver.json; python3 -c "
import json
d=json.load(open('$O/bench_driver.json')); print(d['ms_per_step'], d['roofline']['kernel_us_avg'], d['roofline']['frac']); print({k:(v.get('kernel_us'),v.get('frac')) if isinstance(v,dict) else v for k,v in d.get('other_configs',{}).items()})"
echo "== pmc"
for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
  rm -rf /tmp/pmc_x
  (cd /tmp && timeout 300 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmc_x -o pmc -- python $ROOT/bench.py --steps 12 --warmup 4 --blocks 2 --settle-ms 0 --no-cpu-baseline --no-extras > /dev/null 2>&1)
  f=$(find /tmp/pmc_x -name "*counter_collection*" | head -1)
  [ -n "$f" ] && { echo "--- c2 $PMC"; python $ROOT/tools/pmc_summary.py "$f" jpeg_coeffs; }
done 2>&1 | tee $O/pmc_c2.txt
ls $O
