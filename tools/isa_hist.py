#!/usr/bin/env python3
"""Instruction histogram per kernel from a hipcc -save-temps gfx950 .s file."""
import collections
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/jpeg_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"
want = sys.argv[2] if len(sys.argv) > 2 else None
lines = open(path).read().splitlines()
cur = None
ops = {}
for ln in lines:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = m.group(1)
        ops[cur] = collections.Counter()
        continue
    if ln.startswith(".Lfunc_end"):
        cur = None
    if cur is None:
        continue
    t = ln.strip()
    if not t or t[0] in ".;/" or t.endswith(":"):
        continue
    ops[cur][t.split()[0]] += 1
for k, c in ops.items():
    if want and want not in k:
        continue
    print(k, "total", sum(c.values()))
    groups = collections.Counter()
    for op, n in c.items():
        g = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
             "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        groups[g] += n
    print("  groups:", dict(groups))
    for op, n in c.most_common(40):
        print("   %-28s %d" % (op, n))
