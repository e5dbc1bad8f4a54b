#!/usr/bin/env python3
"""Static cost model of a kernel's hot path from hipcc -S output: skips the bodies of
`s_cbranch_execz` regions (the quantiser's exact-divide slow path) and weights every VALU
opcode by its measured gfx950 issue cost (tools/ubench/form_rate.hip: 2.1 full rate, 4.2 half)."""
import collections, re, sys
path, want = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
FULL = ("v_add_f32_e32 v_sub_f32_e32 v_mul_f32_e32 v_fmaak_f32 v_fmamk_f32 v_and_b32_e32 v_or_b32_e32 v_xor_b32_e32 "
        "v_add_u32_e32 v_sub_u32_e32 v_subrev_u32_e32 v_mov_b32_e32 v_add_u16_e32 v_mul_lo_u16_e32 v_lshrrev_b32_e32 v_lshlrev_b32_e32 "
        "v_add_f32_e64 v_sub_f32_e64 v_mul_f32_e64 v_subrev_f32_e32 v_accvgpr_write_b32 v_accvgpr_read_b32").split()
cur = None; skip_until = None
hot = collections.Counter(); region = 0
sections = collections.OrderedDict()
for li, ln in enumerate(lines):
    m = re.match(r"^(_Z\w+):", ln)
    if m: cur = m.group(1); continue
    if ln.startswith(".Lfunc_end"): cur = None
    if cur is None or want not in cur: continue
    t = ln.split(';')[0].strip() if not ln.strip().startswith(';') else ln.strip()
    if not t or t[0] in ";/" : continue
    if t.endswith(":"):
        if skip_until and t[:-1] == skip_until: skip_until = None
        continue
    if t[0] == ".": continue
    if skip_until: continue
    op = t.split()[0]
    if op == "s_barrier": region += 1
    hot[(region, op)] += 1
    if op.startswith("s_cbranch_"):
        # quantiser slow path (fallthrough body) is skipped; other execz regions are small
        tgt = t.split()[1]
        # only skip when the body contains v_div_scale (look ahead)
        i = li
        body = []
        found = False
        for l2 in lines[i + 1:i + 3000]:
            if l2.split(";")[0].strip() == tgt + ":": found = True; break
            body.append(l2)
        if found and any("v_div_scale" in b for b in body): skip_until = tgt
def cost(op, ln=None):
    if not op.startswith("v_"): return 0.0
    if op in FULL: return 2.1
    if op == "v_fmac_f32_e32": return 2.6
    if op == "v_cndmask_b32_e32" or op == "v_cndmask_b32_e64": return 4.2
    return 4.2
for reg in sorted(set(r for r, _ in hot)):
    ops = {o: n for (r, o), n in hot.items() if r == reg}
    nv = sum(n for o, n in ops.items() if o.startswith("v_"))
    cyc = sum(n * cost(o) for o, n in ops.items())
    print("region %d (after %d barriers): %d instr, %d VALU, modelled VALU cycles %.0f; salu %d, lds %d, vmem %d, waitcnt %d" % (
        reg, reg, sum(ops.values()), nv, cyc, sum(n for o, n in ops.items() if o.startswith("s_") and o != "s_waitcnt"),
        sum(n for o, n in ops.items() if o.startswith("ds_")), sum(n for o, n in ops.items() if o.startswith(("global_", "buffer_"))), ops.get("s_waitcnt", 0)))
    for o, n in sorted(ops.items(), key=lambda kv: -kv[1] * max(cost(kv[0]), 0.5))[:28]:
        print("     %-26s %4d  x %.1f = %6.0f" % (o, n, cost(o), n * cost(o)))
