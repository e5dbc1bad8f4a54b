#!/bin/bash
# Round 2, GPU call A: parity, A/B of the coefficient-kernel variants, unaligned widths, the exit crash of mt_throughput.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2a; mkdir -p $O; export TMPDIR=/tmp
{ rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; } > $O/box.txt 2>&1
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.txt
echo "== A/B"
ab() { # name lib
  for rep in 1 2 3; do
    PIXO_HIP_LIB=${2:+$PWD/$2} timeout 300 python bench.py --steps 400 --warmup 50 --no-cpu-baseline ${AB_ARGS:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-12s %-8s value %9.0f Mpx/s  kernel %7.2f us  pairs med %7.2f min %7.2f  frac %.3f' % ('$1', '${AB_ARGS:-c2}', d['value'], r['kernel_us_avg'], r['kernel_us_event_pairs_median'] or 0, r['kernel_us_event_pairs_min'] or 0, r['frac']))"
  done
}
{ ab default ""; ab r01 pixo_amd/ab_r01.so; ab pk pixo_amd/ab_pk.so; ab plain pixo_amd/ab_plain.so; ab pk_plain pixo_amd/ab_pk_plain.so; ab default ""; ab r01 pixo_amd/ab_r01.so;
  AB_ARGS="--workload c3" ab default ""; AB_ARGS="--workload c3" ab r01 pixo_amd/ab_r01.so; AB_ARGS="--workload c3" ab pk pixo_amd/ab_pk.so;
  AB_ARGS="--workload c2_444" ab default ""; AB_ARGS="--workload c2_444" ab r01 pixo_amd/ab_r01.so; AB_ARGS="--workload c2_444" ab pk pixo_amd/ab_pk.so; } 2>&1 | tee $O/ab.txt
echo "== unaligned"
{ for lib in "" pixo_amd/ab_r01.so pixo_amd/ab_pk.so; do echo "lib ${lib:-default}"; PIXO_HIP_LIB=${lib:+$PWD/$lib} timeout 200 python tools/unaligned_probe.py 2>&1 | tail -6; done; } | tee $O/unaligned.txt
echo "== mt exit crash"
{ SIZE=4096 REPS=6 timeout 200 python -X faulthandler tools/mt_throughput.py; echo "exit code $?";
  echo "--- with PIXO_HIP_KEEP_ON_THREAD_EXIT=1"; PIXO_HIP_KEEP_ON_THREAD_EXIT=1 SIZE=4096 REPS=6 timeout 200 python -X faulthandler tools/mt_throughput.py; echo "exit code $?"; } > $O/mt.txt 2>&1
tail -30 $O/mt.txt
echo "== rocgdb"
SIZE=4096 REPS=6 timeout 400 rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex bt -ex "info threads" -ex "thread apply all bt 12" --args python tools/mt_throughput.py > $O/gdb.txt 2>&1
grep -n "signal\|SIG\|#0\|#1 \|#2 \|#3 \|#4 \|#5 \|#6 \|#7 \|#8 \|exited" $O/gdb.txt | head -60
echo "== rocprof"
for v in default pk; do
  lib=""; [ $v = pk ] && lib=$PWD/pixo_amd/ab_pk.so
  rm -rf /tmp/prof_$v; (cd /tmp && PIXO_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o kt -- python $ROOT/bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $ROOT/$O/bench_under_trace_$v.log 2>&1)
  find /tmp/prof_$v -name "*kernel_stats*" -exec cp {} $O/kernel_stats_$v.csv \;
  head -3 $O/kernel_stats_$v.csv | cut -c1-200
done
ls $O
