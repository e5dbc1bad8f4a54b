cd $GRAFT_REPO_ROOT
python tools/device_time_batch.py 4 4096 4096 2>&1 | tail -2
python tools/device_time_batch.py 64 2048 1088 2>&1 | tail -2
python tools/device_time_batch.py 16 1920 4320 2>&1 | tail -2
python tools/device_time_batch.py 8 4096 2048 2>&1 | tail -2
python tools/device_time_batch.py 32 4096 512 2>&1 | tail -2
