#!/bin/bash
# Round 2, GPU call B: the whole GPU suite with the band/multi tests, the driver's bench command, c4, thread-exit runs.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2b; mkdir -p $O; export TMPDIR=/tmp
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
echo "== bench (driver form)"; timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1 > $O/bench_driver.json; cut -c1-1500 $O/bench_driver.json; tail -3 $O/bench_driver.err
echo "== bench default"; timeout 900 python3 bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c1-700 | tee $O/bench_default.json
echo "== bench c4 (one GPU)"; timeout 900 python3 bench.py --workload c4 --steps 5 --warmup 2 2>$O/bench_c4.err | tail -1 > $O/bench_c4.json; cut -c1-1200 $O/bench_c4.json; tail -5 $O/bench_c4.err
echo "== mt"; { SIZE=4096 REPS=6 timeout 200 python -X faulthandler tools/mt_throughput.py; echo "exit code $?"; } 2>&1 | tail -8 | tee $O/mt.txt
echo "== multi timing"; timeout 300 python tools/multi_timing.py 2>&1 | tail -12 | tee $O/multi_timing.txt
ls $O
