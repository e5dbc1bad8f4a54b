cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest -m gpu -x -q tests/test_gpu_pixels_code.py 2>&1 | grep -E "passed|failed|error|Error" | tail -5
for rep in 1 2 3; do
  PIXO_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/ab_prev.so python tools/device_time.py 2>&1 | tail -1
  python tools/device_time.py 2>&1 | tail -1
done
python tools/device_time_batch.py 64 1920 1080 2>&1 | tail -3
