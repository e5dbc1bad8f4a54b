#!/bin/bash
# Round 2, GPU call D: the single-pass entropy kernels (jpeg_scan_fused.hip): parity suite, kernel times old vs new, whole file.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2d; mkdir -p $O; export TMPDIR=/tmp
echo "== quick sanity (timeout guards a hang in the look-back)"
timeout 120 python -c "
import sys; sys.path.insert(0,'tests')
import synth, oracle_lib as O
from pixo_amd import jpeg
for (w,h,ss) in [(64,64,1),(200,120,1),(1024,520,0),(4096,4096,1)]:
    px = synth.noise(w,h,9)
    o = jpeg.JpegOptions.builder(w,h).quality(80).subsampling(jpeg.Subsampling(ss)).build()
    got = jpeg.encode(px,o); want = O.encode(px, O.make_options(w,h,O.RGB,80,ss))
    print(w,h,ss, len(got), len(want), got==want, flush=True)
" 2>&1 | grep -v amdgpu.ids | tee $O/sanity.txt
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.txt
echo "== whole file timing new vs old"
for mode in new old; do
  [ $mode = old ] && export PIXO_HIP_OLD_ENTROPY=1 || unset PIXO_HIP_OLD_ENTROPY
  echo "-- $mode"; timeout 200 python tools/e2e_device.py 2>&1 | tail -4
  for k in "0 noise" "0 gradient"; do
    n=$(echo $k | tr " " "_")
    rm -rf /tmp/prof_e_${mode}_$n
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e_${mode}_$n -o e -- python $ROOT/tools/encode_loop.py 20 $k 2>&1 | grep "encode()")
    find /tmp/prof_e_${mode}_$n -name "*kernel_stats*" -exec cp {} $O/kernel_stats_encode_${mode}_$n.csv \;
    cut -d, -f1-4 $O/kernel_stats_encode_${mode}_$n.csv | head -12 | cut -c1-150
  done
done 2>&1 | tee $O/timing.txt
unset PIXO_HIP_OLD_ENTROPY
echo "== bench (driver form)"; timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_driver.json; python -c "
import json; d=json.load(open('$O/bench_driver.json')); print(d['ms_per_step'], d['roofline']['frac'], d.get('whole_file'))"
echo "== c4"; timeout 900 python3 bench.py --workload c4 --steps 5 --warmup 2 2>/dev/null | grep '^{' | tail -1 > $O/bench_c4.json; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print(d['ms_per_step'], d['value'])"
