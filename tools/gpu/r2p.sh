#!/bin/bash
# Round 2, GPU call P: the evidence run after the entropy-stage rewrite — parity suite, the driver's bench command, other workloads, rocprofv3 kernel
# statistics and PMC traffic of the coefficient kernel, entropy-stage kernel times (single-pass vs multi-pass), whole files.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2p; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
{ rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; } > $O/box.txt 2>&1
echo "== pytest"; timeout 600 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | grep -v "$F" | tail -14 | tee $O/pytest.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -2 | tee $O/smoke.txt
echo "== bench (driver form)"; timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $O/bench_driver.json; cut -c1-400 $O/bench_driver.json
echo "== bench default"; timeout 900 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 > $O/bench_default.json; cut -c1-300 $O/bench_default.json
for wl in c2_unaligned c2_444 c3 c5; do timeout 600 python3 bench.py --workload $wl --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1; done > $O/bench_other.jsonl; cut -c1-260 $O/bench_other.jsonl
echo "== c4"; timeout 900 python3 bench.py --workload c4 --steps 5 --warmup 2 2>/dev/null | grep '^{' | tail -1 > $O/bench_c4.json; cut -c1-400 $O/bench_c4.json
echo "== rocprof kernel stats (c2)"
rm -rf /tmp/prof_c2; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -o kt -- python $ROOT/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-extras > $ROOT/$O/bench_under_trace.log 2>&1)
find /tmp/prof_c2 -name "*kernel_stats*" -exec cp {} $O/kernel_stats_c2.csv \; ; head -3 $O/kernel_stats_c2.csv | cut -c1-200
echo "== pmc"
for wl in c2 c2_unaligned; do
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
    tag=$(echo $PMC | tr " " "_" | cut -c1-20)
    rm -rf /tmp/pmc_${wl}_${tag}
    (cd /tmp && timeout 300 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmc_${wl}_${tag} -o pmc -- python $ROOT/bench.py --workload $wl --steps 12 --warmup 4 --blocks 2 --settle-ms 0 --no-cpu-baseline --no-extras > /dev/null 2>&1)
    f=$(find /tmp/pmc_${wl}_${tag} -name "*counter_collection*" | head -1)
    [ -n "$f" ] && { echo "--- $wl $PMC"; python $ROOT/tools/pmc_summary.py "$f" jpeg_coeffs; }
  done
done 2>&1 | tee $O/pmc.txt | tail -30
echo "== entropy stage kernels"
for mode in new old; do
  [ $mode = old ] && export PIXO_HIP_OLD_ENTROPY=1 || unset PIXO_HIP_OLD_ENTROPY
  for k in "0 noise" "0 gradient" "1 noise" "1 gradient"; do
    n=$(echo $k | tr " " "_")
    rm -rf /tmp/prof_e_${mode}_$n
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e_${mode}_$n -o e -- python $ROOT/tools/encode_loop.py 20 $k 2>&1 | grep "encode()")
    find /tmp/prof_e_${mode}_$n -name "*kernel_stats*" -exec cp {} $O/kernel_stats_encode_${mode}_$n.csv \;
  done
done 2>&1 | tee $O/encode_loop.txt
unset PIXO_HIP_OLD_ENTROPY
echo "== whole files"
{ timeout 200 python tools/e2e_device.py; timeout 200 python tools/multi_timing.py; SIZE=4096 REPS=6 timeout 200 python tools/mt_throughput.py; echo "mt exit code $?"; timeout 200 python tools/preset2_timing.py | tail -9; } 2>&1 | grep -v "$F" | tee $O/whole_files.txt
ls $O
