cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6w
timeout 1500 python -m pytest -m gpu -x -q tests/test_gpu_pixels_code.py 2>&1 | grep -E "passed|failed|error" | tail -3
for rep in 1 2 3; do
  for v in head "" prio1 prio3 sleep0 sleep4; do
    if [ -z "$v" ]; then python tools/device_time.py 2>&1 | tail -1; else PIXO_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/ab_$v.so python tools/device_time.py 2>&1 | tail -1; fi
  done
done
