cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6o
timeout 1500 python -m pytest -m gpu -x -q tests/test_gpu_pixels_code.py tests/test_gpu_parity.py tests/test_gpu_progressive.py 2>&1 | grep -E "passed|failed|error" | tail -3
bash tools/gpu/call.sh r6o issue pixels_code_noise "pixels_code_kernel" python tools/profile_loop.py noise baseline 20 2>&1 | grep -o '"insts_valu_per_launch": [0-9]*'
bash tools/gpu/call.sh r6o issue pixels_code_gradient "pixels_code_kernel" python tools/profile_loop.py gradient baseline 20 2>&1 | grep -o '"insts_valu_per_launch": [0-9]*'
