cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest -m gpu -x -q tests/test_gpu_pixels_code.py tests/test_gpu_parity.py tests/test_gpu_progressive.py 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do python tools/mt_first_calls.py 3 2>&1 | tail -1 | cut -c1-150; done
for i in 1 2 3 4; do python tools/mt_first_calls.py 6 2>&1 | tail -1 | cut -c1-100; done
for i in 1 2 3; do TS=3,4,3,4,8 python tools/mt_device_files.py noise gradient 2>&1 | tail -2; done
PIXO_HIP_DEBUG=two_kernel_scan python tools/mt_first_calls.py 4 2>&1 | tail -1 | cut -c1-150
python tools/device_time.py 2>&1 | tail -1
python - <<'PY'
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
from pixo_amd import jpeg
print("gate stats (waits, timeouts):", jpeg.dispatch_gate_stats())
PY
