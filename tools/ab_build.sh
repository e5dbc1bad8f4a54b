#!/bin/bash
# Builds kernel variants for A/B timing: tools/ab_build.sh name "<extra hipcc flags>" ...
# -> pixo_amd/ab_<name>.so (same C ABI; select with PIXO_HIP_LIB=...).  Only jpeg_kernels.hip is recompiled per
# variant (the flags are for the coefficient kernel); the other translation units are compiled once into /tmp/pixo_ab_obj.
set -e
cd "$(dirname "$0")/../pixo_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize"
OBJ=/tmp/pixo_ab_obj; mkdir -p $OBJ
for f in jpeg_integer.hip jpeg_entropy.hip jpeg_scan_fused.hip jpeg_trellis.hip png_filter.hip context.cpp scan_job.cpp pieces.cpp progressive.cpp jpeg_api.cpp png_api.cpp bands.cpp jpeg_host.cpp; do
  o=$OBJ/${f%.*}.o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find . ../../include -name '*.h*' -newer $o | head -1)" ]; then /opt/rocm/bin/hipcc $FLAGS -c $f -o $o & fi
done
wait
# (the shipped build's flag for jpeg_kernels.hip, see the Makefile; a variant may override it with its own -mllvm option)
PRELOAD="-mllvm -amdgpu-kernarg-preload-count=14"
while [ $# -ge 2 ]; do
  case "$2" in *NO_PRELOAD*) P="";; *) P="$PRELOAD";; esac
  /opt/rocm/bin/hipcc $FLAGS $P $2 -c jpeg_kernels.hip -o $OBJ/jpeg_kernels_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../ab_$1.so $OBJ/jpeg_kernels_$1.o $OBJ/jpeg_integer.o $OBJ/jpeg_entropy.o $OBJ/jpeg_scan_fused.o $OBJ/jpeg_trellis.o $OBJ/png_filter.o $OBJ/context.o $OBJ/scan_job.o $OBJ/pieces.o $OBJ/progressive.o $OBJ/jpeg_api.o $OBJ/png_api.o $OBJ/bands.o $OBJ/jpeg_host.o
  echo "built ab_$1.so ($2)"
  shift 2
done
