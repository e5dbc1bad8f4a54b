cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6t
timeout 1500 python -m pytest -m gpu -x -q tests/test_gpu_pixels_code.py tests/test_gpu_parity.py tests/test_gpu_progressive.py tests/test_gpu_png.py 2>&1 | grep -E "passed|failed|error" | tail -3
for rep in 1 2 3; do
  PIXO_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/ab_head.so python tools/device_time.py 2>&1 | tail -1
  python tools/device_time.py 2>&1 | tail -1
done
PIXO_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/ab_head.so python tools/device_time.py two 2>&1 | tail -1
python tools/device_time.py two 2>&1 | tail -1
bash tools/gpu/call.sh r6t issue pixels_code_noise "pixels_code_kernel" python tools/profile_loop.py noise baseline 20 2>&1 | grep -o '"insts_valu_per_launch": [0-9]*'
bash tools/gpu/call.sh r6t issue pixels_code_gradient "pixels_code_kernel" python tools/profile_loop.py gradient baseline 20 2>&1 | grep -o '"insts_valu_per_launch": [0-9]*'
bash tools/gpu/call.sh r6t ab c5 2 head base 2>&1 | grep -v "^===="
bash tools/gpu/call.sh r6t bench --gpus 1 --steps 20 --warmup 5 2>&1 | tail -30
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6t/bench_line.json").read())
def find(o, key):
    if isinstance(o, dict):
        for k, v in o.items():
            if k == key: return v
            r = find(v, key)
            if r is not None: return r
    return None
print(json.dumps(find(d, "throughput_by_calling_threads")))
print(json.dumps(find(d, "device_time"))[:1800])
PY
