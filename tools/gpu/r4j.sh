#!/bin/bash
# Round 3, GPU call V: the final evidence run — parity suite, smoke, the driver's bench command, rocprofv3 kernel statistics and PMC
# traffic of the coefficient kernel (c2) and the PNG kernel (c5), entropy-stage kernel times, whole-file timings.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r4j; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
{ rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; } > $O/box.txt 2>&1
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | grep -v "$F" | tail -14 | tee $O/pytest.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -2 | tee $O/smoke.txt
echo "== bench (driver form)"; timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $O/bench_driver.json; cut -c1-400 $O/bench_driver.json
echo "== bench default"; timeout 900 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 > $O/bench_default.json; cut -c1-300 $O/bench_default.json
echo "== rocprof kernel stats (c2)"
rm -rf /tmp/prof_c2; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -o kt -- python $ROOT/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-extras > $ROOT/$O/bench_under_trace.log 2>&1)
find /tmp/prof_c2 -name "*kernel_stats*" -exec cp {} $O/kernel_stats_c2.csv \; ; head -3 $O/kernel_stats_c2.csv | cut -c1-200
echo "== pmc"
for wl in c2 c5; do
  pat=jpeg_coeffs; [ $wl = c5 ] && pat=png_filter
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
    tag=$(echo $PMC | tr " " "_" | cut -c1-20)
    rm -rf /tmp/pmc_${wl}_${tag}
    (cd /tmp && timeout 300 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmc_${wl}_${tag} -o pmc -- python $ROOT/bench.py --workload $wl --steps 12 --warmup 4 --blocks 2 --settle-ms 0 --no-cpu-baseline --no-extras > /dev/null 2>&1)
    f=$(find /tmp/pmc_${wl}_${tag} -name "*counter_collection*" | head -1)
    [ -n "$f" ] && { echo "--- $wl $PMC"; python $ROOT/tools/pmc_summary.py "$f" $pat; }
  done
done 2>&1 | tee $O/pmc.txt | tail -30
echo "== entropy stage kernels"
for k in "0 noise" "0 gradient" "1 noise"; do
  n=$(echo $k | tr " " "_")
  rm -rf /tmp/prof_e_$n
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e_$n -o e -- python $ROOT/tools/encode_loop.py 20 $k 2>&1 | grep "encode()")
  find /tmp/prof_e_$n -name "*kernel_stats*" -exec cp {} $O/kernel_stats_encode_$n.csv \;
done 2>&1 | tee $O/encode_loop.txt
echo "== whole files"
{ timeout 300 python tools/host_pipeline_timing.py; timeout 300 python tools/batch_restart_timing.py; timeout 300 python tools/multi_timing.py; timeout 200 python tools/preset2_timing.py | tail -9; } 2>&1 | grep -v "$F" | tee $O/whole_files.txt
ls $O
echo "== png per strategy"; timeout 300 python tools/png_probe.py 2>&1 | grep -v "$F" | tee $O/png_per_strategy.txt
ls $O
