cd $GRAFT_REPO_ROOT
for i in 1 2; do TS=3,4,3,4,8 python tools/mt_device_files.py noise 2>&1 | tail -1; done
PIXO_HIP_DEBUG=trace TS=3,4,3 python tools/mt_device_files.py noise > gpurun_out/trace_mt.txt 2>&1; tail -1 gpurun_out/trace_mt.txt
grep -n "ms" gpurun_out/trace_mt.txt | awk '{for(i=1;i<=NF;i++) if ($i ~ /^[0-9.]+$/ && $i+0 > 5.0) {print; break}}' | head -40
