#!/bin/bash
# Regenerates every counter profile bench.py reads (profiles/issue_*.json, profiles/traffic_*.json + raw rows) on the tree's library,
# stamped with the library version and the hash of each kernel's sources (tools/profile_meta.py): gpurun -- 'bash tools/gpu/profiles.sh <tag> [set]'
# set: all (default) | coeffs | files | png | tuple
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-profiles}"; SET="${2:-all}"
B="python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --blocks 3 --settle-ms 20 --workload"
run() { bash tools/gpu/call.sh "$TAG" "$@" 2>&1 | tail -3; }
if [ $SET = all ] || [ $SET = coeffs ]; then
  run issue c2 "jpeg_coeffs_kernel<0, 1, false, false>" $B c2
  run traffic c2 "jpeg_coeffs_kernel<0, 1, false, false>" 100663296 $B c2
  run issue c2_444 "jpeg_coeffs_kernel<1, 1, false, true>" $B c2_444
  run traffic c2_444 "jpeg_coeffs_kernel<1, 1, false, true>" 150994944 $B c2_444
  run issue c3 "jpeg_coeffs_kernel<0, 1, false, true>" $B c3
  run traffic c3 "jpeg_coeffs_kernel<0, 1, false, true>" 799211520 $B c3
  run issue c2_unaligned "jpeg_coeffs_kernel<0, 2, false, false>" $B c2_unaligned
  run traffic c2_unaligned "jpeg_coeffs_kernel<0, 2, false, false>" 100638720 $B c2_unaligned
fi
if [ $SET = all ] || [ $SET = png ]; then
  run issue c5 "png_filter_kernel" $B c5
  run traffic c5 "png_filter_kernel" 134221824 $B c5
fi
if [ $SET = all ] || [ $SET = files ]; then
  run issue pixels_code_noise "pixels_code_kernel" python tools/profile_loop.py noise baseline 20
  run traffic pixels_code_noise "pixels_code_kernel" 61481781 python tools/profile_loop.py noise baseline 20
  run issue pixels_code_photo "pixels_code_kernel" python tools/profile_loop.py photo baseline 20
  run traffic pixels_code_photo "pixels_code_kernel" 53044921 python tools/profile_loop.py photo baseline 20
  run issue pixels_code_gradient "pixels_code_kernel" python tools/profile_loop.py gradient baseline 20
  run traffic pixels_code_gradient "pixels_code_kernel" 50654580 python tools/profile_loop.py gradient baseline 20
fi
if [ $SET = all ] || [ $SET = tuple ]; then
  run issue scan_code_noise "scan_code_kernel" python tools/profile_loop.py noise two 20
  run issue scan_code_smooth "scan_code_kernel" python tools/profile_loop.py gradient two 20
  run issue stuff_noise "stuff_fused_kernel" python tools/profile_loop.py noise two 20
  run issue prog_code_noise "prog_code_kernel" python tools/profile_loop.py noise progressive 12
  run traffic prog_code_noise "prog_code_kernel" 61500000 python tools/profile_loop.py noise progressive 12
  run issue trellis "trellis" python tools/profile_loop.py noise preset2 10
fi
ls gpurun_out/$TAG/*.json
