cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest -m gpu -x -q tests/test_gpu_pixels_code.py 2>&1 | grep -E "passed|failed|error|Error" | tail -5
python tools/device_time_batch.py 64 1920 1080 2>&1 | tail -3
python tools/device_time_batch.py 32 4096 512 2>&1 | tail -2
python tools/device_time_batch.py 4 4096 4096 2>&1 | tail -2
python tools/device_time_batch.py 256 640 480 2>&1 | tail -2
python tools/device_time.py 2>&1 | tail -1
