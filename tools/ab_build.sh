#!/bin/bash
# Builds kernel variants for A/B timing: tools/ab_build.sh name "<extra hipcc flags>" ...
# -> pixo_amd/ab_<name>.so (same C ABI; select with PIXO_HIP_LIB=...)
set -e
cd "$(dirname "$0")/../pixo_amd/csrc"
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize $2 -shared -o ../ab_$1.so jpeg_kernels.hip jpeg_integer.hip jpeg_entropy.hip jpeg_scan_fused.hip jpeg_trellis.hip png_filter.hip capi.cpp jpeg_host.cpp
  echo "built ab_$1.so ($2)"
  shift 2
done
