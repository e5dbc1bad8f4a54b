#!/bin/bash
# Round 3, GPU call R: late start for the later groups of scan_code (A/B builds), Bigrams re-measured, stress tools.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; O=gpurun_out/r3r; mkdir -p $O; export TMPDIR=/tmp
F='RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids'
echo "== scan_code stagger A/B (one piece: PIXO_HIP_DEBUG=one_piece; default: pieces)"
for v in new sc10x1 sc10x2 sc10x3 sc9g1 sc8g1 sc9x1 sc9x2 sc8g2; do
  lib=""; [ $v != new ] && lib="$ROOT/pixo_amd/ab_$v.so"
  for mode in one_piece ""; do
    rm -rf /tmp/prof_s
    r=$(cd /tmp && PIXO_HIP_LIB=$lib PIXO_HIP_DEBUG=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $ROOT/tools/encode_loop.py 10 0 noise 2>&1 | grep "encode()" | tail -1 | tr '\n' ' ')
    k=$(python3 - <<PY
import csv,glob
f=glob.glob('/tmp/prof_s/*kernel_stats*')
out=[]
if f:
    for r in csv.DictReader(open(f[0])):
        n=r['Name']
        if 'scan_code' in n or 'stuff_fused' in n: out.append('%s %s x%s' % (n.split('(')[0][-28:], r['AverageNs'][:8], r['Calls']))
print(' | '.join(out))
PY
)
    echo "$v [$mode] $r || $k"
  done
done 2>&1 | tee $O/scan_stagger.txt
echo "== bigrams"; timeout 300 python tools/bigrams_probe.py 2>&1 | grep -v "$F" | tail -12 | tee $O/bigrams.txt
echo "== stress"; timeout 600 python tools/stress_parity.py 120 2>&1 | grep -v "$F" | tail -4 | tee $O/stress_parity.txt
timeout 400 python tools/stress_pieces.py 2>&1 | grep -v "$F" | tail -3 | tee $O/stress_pieces.txt
ls $O
