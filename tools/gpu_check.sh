#!/bin/bash
# One gpurun call: parity tests, smoke, bench, rocprofv3 kernel trace.  Outputs -> gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/rocminfo.txt 2>&1
nproc >> gpurun_out/rocminfo.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/rocminfo.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --steps 200 --warmup 20 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "== rocprof"
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_out -o r01 -- python "$OLDPWD/bench.py" --steps 100 --warmup 10 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof/bench_under_prof.log" 2>&1)
find /tmp/prof_out -name "*stats*" -exec cp {} gpurun_out/prof/ \; 2>/dev/null
ls -la /tmp/prof_out/* 2>/dev/null | head; ls gpurun_out/prof
