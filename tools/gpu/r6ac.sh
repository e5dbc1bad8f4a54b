cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest -m gpu -x -q tests/test_gpu_pixels_code.py tests/test_gpu_parity.py tests/test_gpu_progressive.py 2>&1 | grep -E "passed|failed|error|Error" | tail -5
for rep in 1 2; do
  PIXO_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/ab_head.so python tools/device_time.py two 2>&1 | tail -1
  python tools/device_time.py two 2>&1 | tail -1
  PIXO_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/ab_head.so python tools/device_time_batch.py 2>&1 | tail -2
  python tools/device_time_batch.py 2>&1 | tail -2
done
