#!/bin/bash
# PMC passes for the PNG filter kernel (config 5); outputs -> gpurun_out/pmc_c5/
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_c5; rm -rf /tmp/pmc5
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmc5/p$i -o pmc -- python $ROOT/bench.py --workload c5 --steps 12 --warmup 4 --settle-ms 0 --no-cpu-baseline > "$ROOT/gpurun_out/pmc_c5/pmc$i.log" 2>&1)
  f=$(find /tmp/pmc5/p$i -name "*counter_collection*" | head -1)
  [ -n "$f" ] && python $ROOT/tools/pmc_summary.py "$f" png_filter > gpurun_out/pmc_c5/pmc${i}_summary.txt 2>&1
done
cat gpurun_out/pmc_c5/pmc*_summary.txt
