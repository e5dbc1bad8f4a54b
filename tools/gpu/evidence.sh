#!/bin/bash
# The round's evidence from ONE call: the whole GPU suite, smoke, the default bench line (the driver's command), rocprofv3 --kernel-trace --stats of the
# metric's command from the same box, the PNG counter profiles:   gpurun -- 'bash tools/gpu/evidence.sh <tag>'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-evidence}"
bash tools/gpu/call.sh "$TAG" info -- tests -x -q tests -- smoke -- bench --gpus 1 --steps 20 --warmup 5 -- kstats c2 python bench.py --workload c2 --no-extras --no-cpu-baseline --steps 200 --warmup 50 --blocks 15
bash tools/gpu/profiles.sh "$TAG" png
