cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6o
{
timeout 1200 python -m pytest -m gpu -x -q tests/test_gpu_pixels_code.py 2>&1 | tail -2
for v in r05 base r05 base r05 base; do
  lib=""; [ $v != base ] && lib=$PWD/tools/ab/ab_$v.so
  PIXO_HIP_LIB=$lib python tools/device_time.py 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r6o/out.txt 2>&1
tail -70 gpurun_out/r6o/out.txt
