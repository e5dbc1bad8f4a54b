#!/usr/bin/env python3
"""VALU-issue roofline of a kernel from ONE rocprofv3 --pmc pass (SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE):
    valu_busy = SQ_ACTIVE_INST_VALU x 4 cycles / (1,024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
— the fraction of the kernel's duration during which the vector ALUs were issuing (1.0 = the kernel cannot go faster without
executing fewer or cheaper vector instructions).  usage: issue_profile.py <counter_collection.csv> <kernel substring> <out.json> [label]
Writes {"kernel", "valu_busy", "insts_valu_per_launch", "active_valu_cycles_per_launch" (SQ_ACTIVE_INST_VALU x 4: cycles in which a SIMD's vector
ALU was issuing, summed over the SIMDs), "waves", "launches", "source"}; bench.py divides the active cycles by 1,024 SIMDs x 2.4 GHz x the
kernel time of ITS run: `frac_issue`."""
import collections, csv, json, sys

path, pat, out = sys.argv[1], sys.argv[2], sys.argv[3]
label = sys.argv[4] if len(sys.argv) > 4 else ""
acc = collections.defaultdict(list)
name = None
with open(path) as fh:
    for row in csv.DictReader(fh):
        k = row.get("Kernel_Name", "")
        if pat not in k:
            continue
        name = k.split("(")[0]
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
if not acc:
    sys.exit("no dispatch of a kernel matching %r in %s" % (pat, path))
mean = {c: sum(v) / len(v) for c, v in acc.items()}
cycles = mean["GRBM_GUI_ACTIVE"] / 8.0
d = {"kernel": name, "label": label, "valu_busy": round(mean["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles), 4),
     "insts_valu_per_launch": round(mean["SQ_INSTS_VALU"]), "active_valu_cycles_per_launch": round(mean["SQ_ACTIVE_INST_VALU"] * 4.0),
     "waves": round(mean.get("SQ_WAVES", 0)),
     "cycles_per_xcd": round(cycles), "launches": len(acc["GRBM_GUI_ACTIVE"]),
     "formula": "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)", "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d))
