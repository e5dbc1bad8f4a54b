#!/bin/bash
# Round 3, GPU call J: the pipelined PNG filter kernel — parity, timing against the one-row kernel and two chunk sizes,
# and where the cycles go (SQ busy / wait counters) for the PNG kernel (c5) and the coefficient kernel (c2).
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3j; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== pytest png"; timeout 900 python -m pytest tests/test_gpu_png.py -x -q 2>&1 | grep -v "$F" | tail -8 | tee $O/pytest_png.txt
echo "== c5 timings"
for v in default png_nopipe png_slots512 png_slots2048; do
  lib=""; [ $v != default ] && lib="$ROOT/pixo_amd/ab_$v.so"
  for rep in 1 2; do
    PIXO_HIP_LIB=$lib timeout 300 python3 bench.py --workload c5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d.get('ms_per_step_min'), d['roofline']['kernel_us_avg'], d['roofline']['frac'])"
  done
done 2>&1 | tee $O/c5_variants.txt
echo "== pmc cycles"
for wl in c5 c2; do
  pat=jpeg_coeffs; [ $wl = c5 ] && pat=png_filter
  for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_LDS_BANK_CONFLICT SQ_IFETCH" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM"; do
    tag=$(echo $PMC | tr " " "_" | cut -c1-24)
    rm -rf /tmp/pmc_${wl}_${tag}
    (cd /tmp && timeout 300 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmc_${wl}_${tag} -o pmc -- python $ROOT/bench.py --workload $wl --steps 12 --warmup 4 --blocks 2 --settle-ms 0 --no-cpu-baseline --no-extras > /dev/null 2>&1)
    f=$(find /tmp/pmc_${wl}_${tag} -name "*counter_collection*" | head -1)
    [ -n "$f" ] && { echo "--- $wl $PMC"; python $ROOT/tools/pmc_summary.py "$f" $pat; }
  done
done 2>&1 | tee $O/pmc_cycles.txt
echo "== c5 FETCH/WRITE"
for PMC in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_t_$PMC
  (cd /tmp && timeout 300 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmc_t_$PMC -o pmc -- python $ROOT/bench.py --workload c5 --steps 12 --warmup 4 --blocks 2 --settle-ms 0 --no-cpu-baseline --no-extras > /dev/null 2>&1)
  f=$(find /tmp/pmc_t_$PMC -name "*counter_collection*" | head -1)
  [ -n "$f" ] && { echo "--- c5 $PMC"; python $ROOT/tools/pmc_summary.py "$f" png_filter; }
done 2>&1 | tee $O/pmc_c5_traffic.txt
ls $O
