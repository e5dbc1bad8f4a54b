cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest -m gpu -x -q tests/test_gpu_pixels_code.py tests/test_gpu_parity.py tests/test_gpu_progressive.py 2>&1 | grep -E "passed|failed|error|Error" | tail -5
for i in 1 2 3 4 5 6 7 8; do python tools/mt_first_calls.py 3 2>&1 | tail -1 | cut -c1-60; done
for i in 1 2; do TS=1,2,4,8 python tools/mt_device_files.py gradient 2>&1 | tail -1; done
for rep in 1 2 3; do
  PIXO_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/ab_head.so python tools/device_time.py 2>&1 | tail -1
  python tools/device_time.py 2>&1 | tail -1
done
