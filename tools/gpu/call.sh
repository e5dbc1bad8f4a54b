#!/bin/bash
# The ONE runner for gpurun calls (round 4 replaced 54 one-off r*.sh scripts; what they measured is kept under profiles/).
#   gpurun --timeout T -- 'bash tools/gpu/call.sh <tag> <recipe> [args] [-- <recipe> [args]]...'
# Everything a recipe prints goes to gpurun_out/<tag>/<recipe>_<k>.txt as well.  Recipes:
#   info                      GPU, host cores, cgroup quota
#   smoke                     __graft_entry__.smoke()
#   tests [pytest args]       python -m pytest -m gpu -x -q <args or tests/>
#   bench [bench.py args]     one bench line (summary + the raw line in bench_line.json)
#   ab <workload> <reps> <variant>...   bench.py --no-extras per variant: "base" = the in-tree library, anything else =
#                             tools/ab/ab_<variant>.so (built by tools/ab_build.sh); prints ms_per_step / kernel_us / frac
#   kstats <name> <cmd...>    rocprofv3 --kernel-trace --stats of <cmd>, the kernel_stats csv copied to <tag>/<name>_kernel_stats.csv
#   pmc <name> <filter> <cmd...>   separate rocprofv3 --pmc passes (tools/pmc_summary.py on kernels matching <filter>)
#   issue <name> <filter> <cmd...> one PMC pass -> issue_<name>.json: VALU-busy fraction of the kernel (tools/issue_profile.py)
#   traffic <name> <filter> <algorithmic bytes> <cmd...>   FETCH_SIZE / WRITE_SIZE passes -> traffic_<name>.json + raw rows
#   py <script> [args]        python <script> (tools/*.py probes)
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; export TMPDIR=/tmp
TAG="$1"; shift
O="gpurun_out/$TAG"; mkdir -p "$O"

recipe_info() { rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; grep -m1 "model name" /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4; }
recipe_smoke() { timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5; }
recipe_tests() { if [ $# -eq 0 ]; then set -- tests; fi; timeout 2400 python -m pytest -m gpu -x -q "$@" 2>&1 | tail -25; }
recipe_bench() {
  timeout 1500 python bench.py "$@" 2>"$O/bench_stderr.txt" | grep '^{' | tail -1 > "$O/bench_line.json"
  python3 - "$O/bench_line.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
r = d.get("roofline") or {}
print("value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], d.get("ms_per_step_min"), d.get("ms_per_step_max"), "kernel_us", r.get("kernel_us_avg"), "frac", r.get("frac"))
print("frac by wall", round(r.get("algorithmic_bytes_per_launch", 0) / (d["ms_per_step"] * 1e-3) / 8e12, 4) if r else None)
for k, v in (d.get("other_configs") or {}).items():
    print(" ", k, {a: b for a, b in v.items() if a in ("kernel_us", "frac", "ms_per_step", "ms_per_batch", "value", "error", "leg_wall_s", "bound", "frac_issue")})
for k in ("whole_file", "rccl", "cpu_baseline"):
    if k in d: print(" ", k, json.dumps(d[k])[:400])
PY
  tail -5 "$O/bench_stderr.txt"
}
recipe_ab() {
  wl="$1"; reps="$2"; shift 2
  for rep in $(seq 1 "$reps"); do
    for v in "$@"; do
      lib=""; [ "$v" != base ] && lib="$ROOT/tools/ab/ab_$v.so"
      PIXO_HIP_LIB=$lib timeout 300 python3 bench.py --workload "$wl" --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl $v rep$rep ms_per_step', d['ms_per_step'], 'min', d.get('ms_per_step_min'), 'kernel_us', d['roofline']['kernel_us_avg'], 'frac', d['roofline']['frac'])"
    done
  done
}
# (rocprofv3 runs its command from /tmp: arguments that name files of the repo become absolute paths)
absargs() { ABS=(); for a in "$@"; do if [ -e "$ROOT/$a" ] && [ "${a#/}" = "$a" ]; then ABS+=("$ROOT/$a"); else ABS+=("$a"); fi; done; }
recipe_kstats() {
  name="$1"; shift
  absargs "$@"; set -- "${ABS[@]}"
  rm -rf "/tmp/prof_$name"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$name" -o kt -- "$@" > "$ROOT/$O/${name}_under_trace.log" 2>&1)
  find "/tmp/prof_$name" -name "*kernel_stats*" -exec cp {} "$O/${name}_kernel_stats.csv" \;
  find "/tmp/prof_$name" -name "*kernel_trace*" -exec cp {} "$O/${name}_kernel_trace.csv" \;
  head -12 "$O/${name}_kernel_stats.csv"
}
recipe_pmc() {
  name="$1"; filt="$2"; shift 2
  absargs "$@"; set -- "${ABS[@]}"
  i=0
  for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
             "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
    i=$((i+1)); rm -rf "/tmp/pmc_${name}_$i"
    (cd /tmp && timeout 400 rocprofv3 --pmc $PMC --output-format csv -d "/tmp/pmc_${name}_$i" -o pmc -- "$@" > "$ROOT/$O/${name}_pmc$i.log" 2>&1)
    f=$(find "/tmp/pmc_${name}_$i" -name "*counter_collection*" | head -1)
    [ -n "$f" ] && python "$ROOT/tools/pmc_summary.py" "$f" "$filt" 2>&1 | tee "$O/${name}_pmc${i}_summary.txt"
  done
}
# issue <name> <kernel substring> <cmd...>: one PMC pass -> gpurun_out/<tag>/issue_<name>.json (tools/issue_profile.py)
recipe_issue() {
  name="$1"; filt="$2"; shift 2
  absargs "$@"; set -- "${ABS[@]}"
  rm -rf "/tmp/issue_$name"
  (cd /tmp && timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "/tmp/issue_$name" -o pmc -- "$@" > "$ROOT/$O/issue_${name}.log" 2>&1)
  f=$(find "/tmp/issue_$name" -name "*counter_collection*" | head -1)
  # (the raw counter rows of the kernels matching the filter travel with the summary: profiles/issue_<name>_raw.csv)
  if [ -n "$f" ]; then python "$ROOT/tools/issue_profile.py" "$f" "$filt" "$O/issue_$name.json" "$name"; (head -1 "$f"; grep -F "$filt" "$f") > "$O/issue_${name}_raw.csv"; else tail -5 "$O/issue_${name}.log"; fi
}
# traffic <name> <kernel substring> <algorithmic bytes> <cmd...>: HBM bytes per launch of one kernel — separate --pmc passes for
# FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md: TCC counters do not fit one pass), FETCH_SIZE x 2 on gfx950 —
# -> gpurun_out/<tag>/traffic_<name>.json + the raw counter rows traffic_<name>_raw.csv
recipe_traffic() {
  name="$1"; filt="$2"; alg="$3"; shift 3
  absargs "$@"; set -- "${ABS[@]}"
  : > "$O/traffic_${name}_raw.csv"
  for PMC in FETCH_SIZE WRITE_SIZE; do
    rm -rf "/tmp/traffic_${name}_$PMC"
    (cd /tmp && timeout 400 rocprofv3 --pmc $PMC --output-format csv -d "/tmp/traffic_${name}_$PMC" -o pmc -- "$@" > "$ROOT/$O/traffic_${name}_$PMC.log" 2>&1)
    f=$(find "/tmp/traffic_${name}_$PMC" -name "*counter_collection*" | head -1)
    [ -n "$f" ] && { [ -s "$O/traffic_${name}_raw.csv" ] || head -1 "$f" > "$O/traffic_${name}_raw.csv"; grep -F "$filt" "$f" >> "$O/traffic_${name}_raw.csv"; }
  done
  python3 - "$O/traffic_${name}_raw.csv" "$O/traffic_$name.json" "$alg" "$name" <<'PY'
import collections, csv, json, sys
acc = collections.defaultdict(list)
kern = None
for row in csv.DictReader(open(sys.argv[1])):
    acc[row["Counter_Name"]].append(float(row["Counter_Value"])); kern = row["Kernel_Name"].split("(")[0]
rd = sum(acc["FETCH_SIZE"]) / max(1, len(acc["FETCH_SIZE"])) * 1024 * 2   # KiB, x 2: the gfx950 correction for wide streaming reads
wr = sum(acc["WRITE_SIZE"]) / max(1, len(acc["WRITE_SIZE"])) * 1024
alg = int(sys.argv[3])
d = {"kernel": kern, "label": sys.argv[4], "hbm_bytes_per_launch": round(rd + wr), "read_bytes": round(rd), "write_bytes": round(wr),
     "launches": [len(acc["FETCH_SIZE"]), len(acc["WRITE_SIZE"])], "algorithmic_bytes": alg, "over_algorithmic": round((rd + wr) / alg, 4) if alg else None,
     "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (KiB; FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md HBM section); raw rows: traffic_%s_raw.csv" % sys.argv[4]}
json.dump(d, open(sys.argv[2], "w"), indent=1); print(json.dumps(d))
PY
  python3 "$ROOT/tools/profile_meta.py" --stamp "$O/traffic_$name.json" > /dev/null   # library version + hash of the kernel's sources (staleness guard of bench.py)
}
recipe_py() { timeout 900 python "$@" 2>&1 | tail -60; }

while [ $# -gt 0 ]; do
  r="$1"; shift
  args=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
  [ $# -gt 0 ] && shift
  echo "==== $r ${args[*]:-}"
  n=$(ls "$O" | grep -c "^${r}_") || true
  "recipe_$r" ${args[@]+"${args[@]}"} 2>&1 | tee "$O/${r}_${n}.txt"
done
