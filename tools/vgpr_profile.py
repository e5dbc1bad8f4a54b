#!/usr/bin/env python3
"""Max VGPR index referenced per chunk of a kernel's asm (finds where pressure peaks)."""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
step = int(sys.argv[3]) if len(sys.argv) > 3 else 80
s = open(path).read()
i = s.index(pat + ":")
body = s[i:]
body = body[:body.index(".Lfunc_end")]
lines = body.splitlines()
for i in range(0, len(lines), step):
    chunk = lines[i:i + step]
    regs = [int(x) for l in chunk for x in re.findall(r"\bv(\d+)\b", l)]
    rng = [int(b) for l in chunk for a, b in re.findall(r"v\[(\d+):(\d+)\]", l)]
    tags = [l.strip().split()[0] for l in chunk if "s_barrier" in l or "ds_write_b128" in l]
    print(i, max(regs + rng + [0]), " ".join(sorted(set(tags))))
