cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6f
{
python tools/device_time_batch.py 2>&1 | grep -v amdgpu.ids
timeout 2400 python -m pytest -m gpu -x -q tests 2>&1 | tail -15
} > gpurun_out/r6f/out.txt 2>&1
tail -30 gpurun_out/r6f/out.txt
