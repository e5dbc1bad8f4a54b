cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6x
for i in 1 2 3 4 5 6; do TS=3,4,3,4 python tools/mt_device_files.py noise 2>&1 | tail -1; done
PIXO_HIP_DEBUG=trace TS=4,4,4 python tools/mt_device_files.py noise > gpurun_out/r6x/trace.txt 2>&1; tail -1 gpurun_out/r6x/trace.txt; grep -c . gpurun_out/r6x/trace.txt
