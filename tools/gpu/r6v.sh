cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6v
timeout 1500 python -m pytest -m gpu -x -q tests/test_gpu_pixels_code.py 2>&1 | grep -E "passed|failed|error" | tail -3
for rep in 1 2 3; do
  PIXO_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/ab_head.so python tools/device_time.py 2>&1 | tail -1
  python tools/device_time.py 2>&1 | tail -1
done
