cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6u
PIXO_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/ab_timeline.so python tools/pixels_code_timeline.py noise,gradient 2>&1 | awk '/rep 7/{p=1} /rep 6/{p=0} p' | tee gpurun_out/r6u/timeline.txt
python tools/mt_device_files.py noise gradient 2>&1 | tail -2
python tools/mt_device_files.py noise --sw direct_stores 2>&1 | tail -1
python tools/mt_device_files.py noise --sw one_piece 2>&1 | tail -1
python tools/mt_device_files.py noise --sw two_kernel_scan 2>&1 | tail -1
