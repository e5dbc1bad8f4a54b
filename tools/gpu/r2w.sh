#!/bin/bash
# GPU call W: growing pieces for a medium scan (4096x4096): schedules against one piece, whole-file time into pinned memory
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r2w; mkdir -p $O; export TMPDIR=/tmp
for sched in "1,3" "1,5" "1,2,5" "1,2,6" "1,3"; do
  PIXO_HIP_PIECE_SCHEDULE=$sched timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); w=d.get('whole_file',{}); print('schedule $sched: whole_file', w.get('ms_per_image'), w.get('ms_per_image_min'), 'bytes-path', w.get('ms_per_image_as_python_bytes'), 'kernel', d['ms_per_step'])"
done 2>&1 | tee $O/timing.txt
