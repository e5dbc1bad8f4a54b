cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
  PIXO_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/ab_nodcearly.so python tools/device_time.py 2>&1 | tail -1
  python tools/device_time.py 2>&1 | tail -1
done
PIXO_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/ab_nodcearly.so python tools/device_time_batch.py 64 1920 1080 2>&1 | tail -3
python tools/device_time_batch.py 64 1920 1080 2>&1 | tail -3
