#!/bin/bash
# Round 2, GPU call C: unaligned rows (x4 funnel loads) with rotating buffers + PMC traffic, 4:4:4 A/B, c4 bench line.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2c; mkdir -p $O; export TMPDIR=/tmp
q() { PIXO_HIP_LIB=${2:+$PWD/$2} timeout 300 python bench.py --workload $1 --steps 100 --warmup 20 --blocks 9 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-14s %-10s ms/step %.5f (min %.5f max %.5f)  kernel %7.2f us  frac %.3f' % ('$1', '${2:-default}', d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], r['kernel_us_avg'], r['frac']))"; }
echo "== pytest (coefficient tests)"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "coefficients or edge or golden or every_rgb" 2>&1 | tail -3
{ q c2_unaligned ""; q c2_unaligned pixo_amd/ab_r01.so; q c2 ""; q c2_unaligned ""; q c2_444 ""; q c2_444 pixo_amd/ab_r01.so; q c2_444 pixo_amd/ab_plain.so; q c2_444 ""; q c2_444 pixo_amd/ab_r01.so; q c3 ""; } 2>&1 | tee $O/ab.txt
echo "== pmc"
for wl in c2 c2_unaligned c2_444; do
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"; do
    tag=$(echo $PMC | tr " " "_" | cut -c1-20)
    rm -rf /tmp/pmc_$wl_$tag
    (cd /tmp && timeout 300 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmc_${wl}_$tag -o pmc -- python $ROOT/bench.py --workload $wl --steps 12 --warmup 4 --blocks 2 --settle-ms 0 --no-cpu-baseline --no-extras > /dev/null 2>&1)
    f=$(find /tmp/pmc_${wl}_$tag -name "*counter_collection*" | head -1)
    [ -n "$f" ] && { echo "--- $wl $PMC"; python $ROOT/tools/pmc_summary.py "$f" jpeg_coeffs; }
  done
done 2>&1 | tee $O/pmc.txt
echo "== c4"; timeout 900 python3 bench.py --workload c4 --steps 5 --warmup 2 2>$O/bench_c4.err | grep '^{' | tail -1 > $O/bench_c4.json; cut -c1-1500 $O/bench_c4.json; grep -v amdgpu.ids $O/bench_c4.err | tail -5
