#!/bin/bash
# One gpurun call: parity tests, smoke, bench, rocprofv3 kernel trace + PMC passes.
# Outputs -> gpurun_out/.   usage: tools/gpu_check.sh [quick|full|prof]
set -u
MODE="${1:-full}"
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
{ rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; grep -m1 "model name" /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max 2>/dev/null; } > gpurun_out/rocminfo.txt 2>&1
if [ "$MODE" != "prof" ]; then
  echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
fi
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench.log
if [ "$MODE" = "quick" ]; then exit 0; fi
echo "== extra workloads"
for wl in c2_444 c3 c1; do timeout 300 python bench.py --workload $wl --steps 400 --no-cpu-baseline 2>&1 | tail -1; done | tee gpurun_out/bench_extra.log
echo "== rocprof kernel trace"
rm -rf /tmp/prof_out gpurun_out/prof && mkdir -p gpurun_out/prof
BENCH="python $ROOT/bench.py --steps 1000 --warmup 100 --no-cpu-baseline"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out/trace -o kt -- $BENCH > "$ROOT/gpurun_out/prof/bench_under_trace.log" 2>&1)
find /tmp/prof_out/trace -name "*kernel_stats*" -exec cp {} gpurun_out/prof/ \;
find /tmp/prof_out/trace -name "*kernel_trace*" -exec sh -c 'head -400 "$1" > gpurun_out/prof/kernel_trace_head.csv' _ {} \;
rocprofv3 -L 2>/dev/null | grep -oE "(SQ|TCC|TCP|GRBM|TA)_[A-Z0-9_]+" | sort -u > gpurun_out/prof/counters_available.txt
echo "== rocprof pmc"
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $PMC --output-format csv -d /tmp/prof_out/pmc$i -o pmc -- python $ROOT/bench.py --steps 12 --warmup 4 --settle-ms 0 --no-cpu-baseline > "$ROOT/gpurun_out/prof/pmc$i.log" 2>&1)
  f=$(find /tmp/prof_out/pmc$i -name "*counter_collection*" | head -1)
  [ -n "$f" ] && python $ROOT/tools/pmc_summary.py "$f" jpeg_coeffs > gpurun_out/prof/pmc${i}_summary.txt 2>&1
done
echo "== PNG (c5) and entropy stage"
mkdir -p gpurun_out/extra
timeout 300 python bench.py --workload c5 2>/dev/null | tail -1 > gpurun_out/extra/bench_c5.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out/c5 -o c5 -- python $ROOT/bench.py --workload c5 --steps 400 --no-cpu-baseline > /dev/null 2>&1)
find /tmp/prof_out/c5 -name "*kernel_stats*" -exec cp {} gpurun_out/extra/kernel_stats_c5.csv \;
for k in "0 noise" "1 noise" "0 gradient"; do
  n=$(echo $k | tr " " "_")
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out/e_$n -o e -- python $ROOT/tools/encode_loop.py 10 $k 2>&1 | grep "encode()" > $ROOT/gpurun_out/extra/encode_loop_$n.txt)
  find /tmp/prof_out/e_$n -name "*kernel_stats*" -exec cp {} gpurun_out/extra/kernel_stats_encode_$n.csv \;
done
timeout 120 python tools/e2e_timing.py 2>&1 | tail -3 > gpurun_out/extra/e2e_timing.txt
timeout 120 python tools/e2e_device.py 2>&1 | tail -4 > gpurun_out/extra/e2e_device.txt
timeout 120 python tools/png_probe.py 2>&1 | tail -8 > gpurun_out/extra/png_probe.txt
timeout 120 python tools/preset2_timing.py 2>&1 | tail -9 > gpurun_out/extra/preset2_timing.txt
timeout 120 python tools/mt_throughput.py 2>&1 | tail -5 > gpurun_out/extra/mt_throughput.txt
timeout 60 python tools/warmup_probe.py 2>&1 | tail -2 > gpurun_out/extra/warmup_probe.txt
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out/p2 -o p2 -- python $ROOT/tools/preset2_timing.py > /dev/null 2>&1)
find /tmp/prof_out/p2 -name "*kernel_stats*" -exec cp {} gpurun_out/extra/kernel_stats_preset2.csv \;
bash tools/pmc_c5.sh > gpurun_out/extra/pmc_c5.txt 2>&1
ls gpurun_out/prof gpurun_out/extra
