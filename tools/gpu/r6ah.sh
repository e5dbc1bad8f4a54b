cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest -m gpu -x -q tests 2>&1 | grep -E "passed|failed|error|Error" | tail -5
python tools/device_time_batch.py 64 1920 1080 2>&1 | tail -3
timeout 300 python tools/stress_parity.py 120 811 fused 2>&1 | tail -2
