#!/usr/bin/env python3
"""VALU-issue roofline of a kernel from ONE rocprofv3 --pmc pass (SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE):
    valu_busy = SQ_ACTIVE_INST_VALU x 4 cycles / (1,024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
— the fraction of the kernel's duration during which the vector ALUs were issuing (1.0 = the kernel cannot go faster without
executing fewer or cheaper vector instructions).  usage: issue_profile.py <counter_collection.csv> <kernel substring> <out.json> [label]
Writes {"kernel", "valu_busy", "insts_valu_per_launch", "active_valu_cycles_per_launch" (SQ_ACTIVE_INST_VALU x 4: cycles in which a SIMD's vector
ALU was issuing, summed over the SIMDs), "waves", "launches", "engine_clock_hz_measured" (GRBM_GUI_ACTIVE / 8 over the dispatch's duration),
"library_version", "kernel_source_hash" (tools/profile_meta.py: bench.py reports counters of another build as stale), "source"}; bench.py
divides the active cycles by 1,024 SIMDs x the measured clock x the kernel time of ITS run: `frac_issue`."""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import profile_meta

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
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            acc["_duration_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
if not acc:
    sys.exit("no dispatch of a kernel matching %r in %s" % (pat, path))
mean = {c: sum(v) / len(v) for c, v in acc.items()}
cycles = mean["GRBM_GUI_ACTIVE"] / 8.0
d = {"kernel": name, "label": label, "valu_busy": round(mean["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles), 4),
     "insts_valu_per_launch": round(mean["SQ_INSTS_VALU"]), "active_valu_cycles_per_launch": round(mean["SQ_ACTIVE_INST_VALU"] * 4.0),
     "waves": round(mean.get("SQ_WAVES", 0)),
     "cycles_per_xcd": round(cycles), "launches": len(acc["GRBM_GUI_ACTIVE"]),
     "formula": "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)", "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"}
# The engine clock DURING the profiled launches: busy cycles per XCD over the dispatch's duration under the counters (which is longer
# than an unprofiled launch: the counters perturb the kernel — valu_busy is a fraction of THAT duration; bench.py's frac_issue puts the
# same active cycles over the kernel time of its own, unprofiled run at this clock).
if acc.get("_duration_ns"):
    dur = sum(acc["_duration_ns"]) / len(acc["_duration_ns"])
    d["duration_ns_under_counters"] = round(dur)
    # (GRBM_GUI_ACTIVE's window is ~7 us longer than the dispatch: for a launch of 20 us the quotient says 3.1 GHz, for one of 150 us 2.36 — a
    # usable clock only for long launches; bench.py measures the loaded clock in its own run, pixo_hip_debug_engine_clock)
    d["grbm_cycles_over_duration_hz"] = round(cycles / (dur * 1e-9))
    if dur >= 100e3:
        d["engine_clock_hz_measured"] = round(cycles / (dur * 1e-9))
d.update(profile_meta.meta(label))
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d))
