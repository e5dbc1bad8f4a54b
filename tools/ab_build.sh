#!/bin/bash
# Builds kernel variants for A/B timing: tools/ab_build.sh name "<extra hipcc flags>" ...
# -> tools/ab/ab_<name>.so (same C ABI; select with PIXO_HIP_LIB=...; kept OUT of the package directory and out of git —
# delete the variants after the call that measured them: everything under tools/ab/ is pushed to the GPU box).  Only ONE kernel file is recompiled per variant
# (jpeg_kernels.hip, or the one named by AB_SRC=png_filter.hip); the other translation units are compiled once into
# /tmp/pixo_ab_obj.
set -e
cd "$(dirname "$0")/../pixo_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize"
OBJ=/tmp/pixo_ab_obj; mkdir -p $OBJ ../../tools/ab
SRC=${AB_SRC:-jpeg_kernels.hip}; BASE=${SRC%.*}
for f in jpeg_kernels.hip jpeg_pixels_code.hip jpeg_integer.hip jpeg_entropy.hip jpeg_scan_fused.hip jpeg_trellis.hip png_filter.hip stream_copy.hip context.cpp dispatch_gate.cpp scan_job.cpp pieces.cpp progressive.cpp jpeg_api.cpp png_api.cpp bands.cpp jpeg_host.cpp; do
  o=$OBJ/${f%.*}.o
  K=""; { [ $f = jpeg_kernels.hip ] || [ $f = jpeg_pixels_code.hip ]; } && K="-mllvm -amdgpu-kernarg-preload-count=14"
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find . ../../include -name '*.h*' -newer $o | head -1)" ]; then /opt/rocm/bin/hipcc $FLAGS $K -c $f -o $o & fi
done
wait
# (the shipped build's flag for jpeg_kernels.hip, see the Makefile; a variant may override it with its own -mllvm option)
PRELOAD="-mllvm -amdgpu-kernarg-preload-count=14"
while [ $# -ge 2 ]; do
  case "$2" in *NO_PRELOAD*) P="";; *) P="$PRELOAD";; esac
  { [ $SRC = jpeg_kernels.hip ] || [ $SRC = jpeg_pixels_code.hip ]; } || P=""
  /opt/rocm/bin/hipcc $FLAGS $P $2 -c $SRC -o $OBJ/${BASE}_$1.o
  OBJS=""; for b in jpeg_kernels jpeg_pixels_code jpeg_integer jpeg_entropy jpeg_scan_fused jpeg_trellis png_filter stream_copy; do
    if [ $b = $BASE ]; then OBJS="$OBJS $OBJ/${b}_$1.o"; else OBJS="$OBJS $OBJ/$b.o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/ab/ab_$1.so $OBJS $OBJ/context.o $OBJ/dispatch_gate.o $OBJ/scan_job.o $OBJ/pieces.o $OBJ/progressive.o $OBJ/jpeg_api.o $OBJ/png_api.o $OBJ/bands.o $OBJ/jpeg_host.o
  echo "built ab_$1.so ($2)"
  shift 2
done
