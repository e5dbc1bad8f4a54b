cd $GRAFT_REPO_ROOT
H=$GRAFT_REPO_ROOT/tools/ab/ab_head.so
echo "## device time per 4096x4096 file (tools/device_time.py): library of commit 38d95a8 (start of the session) | final tree"
for ss in 420 444 gray; do for lib in "$H" ""; do echo -n "$ss  "; PIXO_HIP_LIB=$lib SS=$ss python tools/device_time.py 2>&1 | tail -1; done; done
echo "## q = 100 (groups of several rounds)"
for lib in "$H" ""; do PIXO_HIP_LIB=$lib Q=100 python tools/device_time.py 2>&1 | tail -1; done
echo "## 64 x 1920x1080 batch, device time (tools/device_time_batch.py; head: fused = switch fused_batch, default = two)"
for lib in "$H" ""; do PIXO_HIP_LIB=$lib python tools/device_time_batch.py 64 1920 1080 2>&1 | tail -3; done
echo "## whole files, wall us (device pixels -> pinned buffer): optimised tables (tools/preset1_timing.py: head has no fused form, both columns = tuple path)"
for lib in "$H" ""; do PIXO_HIP_LIB=$lib python tools/preset1_timing.py 4096 2>&1 | tail -6 | head -3; done
echo "## large scans (tools/large_scan_paths.py)"
for lib in "$H" ""; do PIXO_HIP_LIB=$lib python tools/large_scan_paths.py 2>&1 | grep -v optimised | grep "4096x4096\|8192x8192" ; done
