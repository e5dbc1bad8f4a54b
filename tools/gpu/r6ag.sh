cd $GRAFT_REPO_ROOT
python tools/device_time_batch.py 64 2048 1088 2>&1 | tail -1
python tools/device_time_batch.py 32 1920 1080 2>&1 | tail -1
python tools/device_time_batch.py 64 4096 544 2>&1 | tail -1
python tools/device_time_batch.py 64 2048 1024 2>&1 | tail -1
python tools/device_time_batch.py 128 2048 512 2>&1 | tail -1
python tools/device_time_batch.py 16 2048 4096 2>&1 | tail -1
