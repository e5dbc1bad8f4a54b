cd $GRAFT_REPO_ROOT
python tools/device_time_tall.py gradient 2>&1 | tail -4
python tools/device_time_tall.py photo 2>&1 | tail -4
