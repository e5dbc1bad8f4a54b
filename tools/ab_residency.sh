#!/bin/bash
# C2 with fewer workgroups resident per CU (unused dynamic LDS) and with/without raised phase-A priority
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for pad in 0 4096 8192 12288 16384 24576; do for prio in 0 1; do
  echo -n "lds_pad=$pad prio_a=$prio: "
  PIXO_HIP_LDS_PAD=$pad PIXO_HIP_PRIO_A=$prio python bench.py --no-cpu-baseline --steps 400 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_us_avg'])"
done; done
